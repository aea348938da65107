// Scene sources for upload: the reference's built-in scene and the synthetic benchmark scene.
#pragma once

#include <vector>

#include "Math/kl_render.h"

namespace kl {

void builtin_scene(std::vector<kl_sphere> &sph, std::vector<kl_quad> &quads, std::vector<kl_material> &mats);
void synth_counts(const kl_synth_params &sp, uint32_t &ns, uint32_t &nq, uint32_t &nt, uint32_t &nm);
void synth_fill(const kl_synth_params &sp, kl_sphere *sph, kl_quad *quads, kl_tri *tris, kl_material *mats);

}  // namespace kl
