/* The C ABI lives next to the reference's include/Math headers; this file only forwards. */
#include "Math/kl_render.h"
