// One-call evaluation render for non-Python hosts ("optionally tp_render_fused_eval"): Graph.render in
// eval / val mode (reference model/nerf_adapt_st_gan.py:547-631 with mode != 'train') = tp_raygen + tp_mlp_fwd +
// tp_composite_fwd enqueued on one stream with the intermediates in a caller-provided workspace.  No new arithmetic:
// the three kernels are the ones the Python mirror launches, so the outputs are bit-identical to it.
#include "tp_common.h"

namespace {
size_t align256(size_t x) { return (x + 255) / 256 * 256; }
struct Layout { size_t center, ray, depth, rgb, density, uncert, mlp_ws, ray_bias, total; };
Layout layout(int64_t B, int64_t R, int64_t N) {
  Layout l;
  size_t o = 0;
  l.center = o; o += align256((size_t)B * R * 3 * 4);
  l.ray = o;    o += align256((size_t)B * R * 3 * 4);
  l.depth = o;  o += align256((size_t)B * R * N * 4);
  l.rgb = o;    o += align256((size_t)B * R * N * 6 * 4);
  l.density = o; o += align256((size_t)B * R * N * 2 * 4);
  l.uncert = o; o += align256((size_t)B * R * N * 4);
  l.mlp_ws = o; o += align256(tp_mlp_workspace_bytes(B * R * N));
  l.ray_bias = o; o += align256(tp_mlp_ray_bias_bytes((int)B, (int)R));
  l.total = o;
  return l;
}
}  // namespace

extern "C" size_t tp_render_eval_workspace_bytes(int B, int R, int N) { return layout(B, R, N).total; }

extern "C" int tp_render_eval(const tp_render_eval_args* a, tp_stream_t stream) {
  TP_REQUIRE(a && a->workspace && a->out_ray && a->packed && a->lat_trans && a->lat_light, "null pointer");
  TP_REQUIRE(a->raygen.B > 0 && a->raygen.R > 0 && a->raygen.N > 0, "bad sizes");
  const int B = a->raygen.B, R = a->raygen.R, N = a->raygen.N;
  const Layout l = layout(B, R, N);
  char* ws = (char*)a->workspace;
  tp_raygen_args rg = a->raygen;
  rg.center = (float*)(ws + l.center); rg.ray = (float*)(ws + l.ray); rg.depth = (float*)(ws + l.depth);
  rg.near = nullptr; rg.far = nullptr;
  if (int rc = tp_raygen(&rg, stream)) return rc;
  tp_mlp_fwd_args m = {};
  m.packed = a->packed;
  m.center = rg.center; m.ray = rg.ray; m.depth = rg.depth;
  m.lat_trans = a->lat_trans; m.lat_light = a->lat_light;
  m.B = B; m.R = R; m.N = N;
  m.rgb = (float*)(ws + l.rgb); m.density = (float*)(ws + l.density); m.uncert = (float*)(ws + l.uncert);
  m.saved = nullptr; m.workspace = ws + l.mlp_ws; m.precision = a->precision; m.status = a->status;
  if (a->packed_ray_bias) {
    TP_REQUIRE(a->precision == TP_MLP_F16X3 && N % 128 == 0, "tp_render_eval: packed_ray_bias needs TP_MLP_F16X3 and N % 128 == 0");
    m.ray_bias = (float*)(ws + l.ray_bias);
  }
  if (int rc = tp_mlp_fwd(&m, stream)) return rc;
  tp_composite_args c = {};
  c.ray = rg.ray; c.rgb = m.rgb; c.density = m.density; c.depth = rg.depth; c.uncert = m.uncert;
  c.n = (int64_t)B * R; c.N = N; c.min_uncert = a->min_uncert;
  c.out_ray = a->out_ray; c.alpha_static = a->alpha_static; c.alpha_transient = a->alpha_transient; c.prob = nullptr;
  return tp_composite_fwd(&c, stream);
}
