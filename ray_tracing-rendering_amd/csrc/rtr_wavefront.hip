/*
 * rtr_wavefront.hip -- the wavefront pipeline's translation unit: stage kernels and their host driver
 * (rt_wavefront.h).
 */
#include "rt_wavefront.h"
