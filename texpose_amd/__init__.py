"""texpose_amd: MI355X-native (gfx950) ray-marching path for TexPose-style neural texture rendering.

Only the hot path of SURVEY.md section 8 lives here: ray generation, stratified sampling, the
static/transient/light MLP, the per-ray composite and the patch gather, as hand-written HIP
kernels behind a C ABI (include/texpose_amd.h), plus the Python mirror of the reference's
Graph / NeRF / RaySampler interface that makes it a drop-in for that path.
"""
__version__ = "0.1.0"
