/* forwards to the host renderer mirror (same role as the reference header of this name) */
#include "../rtr_renderer.h"
