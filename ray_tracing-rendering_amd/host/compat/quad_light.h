/* forwards to the host scene-description API (same role as the reference header of this name) */
#include "../rtr_scene_api.h"
