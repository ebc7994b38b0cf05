// dvbt_hip.hip -- C ABI (include/dvbt_hip.h) over the gfx950 kernels.  There is no CPU path:
// every entry point fails with DVBT_ERR_NO_DEVICE when no HIP device is usable.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <deque>
#include <algorithm>
#include "../../include/dvbt_hip.h"
#include "dvbt_tables.hpp"
#include "ts_ring.hpp"
#include "k_frontend.hpp"
#include "k_drift.hpp"
#include "k_backend.hpp"
#include "k_symbol8k.hpp"
#include "k_symbol2k.hpp"
#include "k_resample.hpp"
#include "k_viterbi3.hpp"
#include "k_soft4.hpp"

using namespace dvbt;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(DVBT_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define HIPCHKV(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); } } while (0)

extern "C" const char *dvbt_last_error(void) { return g_err.c_str(); }
extern "C" const char *dvbt_version(void) { return "dvbt_hip 0.2 (gfx950)"; }
extern "C" int dvbt_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
extern "C" void *dvbt_device_malloc(size_t bytes) { void *p = nullptr; if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { g_err = "hipMalloc failed"; return nullptr; } return p; }
extern "C" void dvbt_device_free(void *p) { if (p) (void)hipFree(p); }
extern "C" int dvbt_copy_to_device(void *dst, const void *src, size_t bytes) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return DVBT_OK; }
extern "C" int dvbt_copy_to_host(void *dst, const void *src, size_t bytes) { HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return DVBT_OK; }
extern "C" int dvbt_synchronize(void *stream) { if (stream) HIPCHK(hipStreamSynchronize((hipStream_t)stream)); else HIPCHK(hipDeviceSynchronize()); return DVBT_OK; }
static int need_device()
{
  if (dvbt_device_count() <= 0) return fail(DVBT_ERR_NO_DEVICE, "no HIP device visible: libdvbt_hip has no CPU fallback");
  return DVBT_OK;
}

// page-lock a host buffer the caller will hand to dvbt_<blk>_work / dvbt_rx_stream_push again and again (a GNU Radio block's input / output buffer): the DMA
// engines then move the items straight from / to it (~50 GB/s, asynchronously) instead of through the handle's pinned staging (one host memcpy each way)
extern "C" int dvbt_host_register(void *p, size_t bytes) { if (!p || !bytes) return fail(DVBT_ERR_INVALID, "null argument"); int r = need_device(); if (r) return r; HIPCHK(hipHostRegister(p, bytes, hipHostRegisterDefault)); return DVBT_OK; }
extern "C" int dvbt_host_unregister(void *p) { if (!p) return fail(DVBT_ERR_INVALID, "null argument"); HIPCHK(hipHostUnregister(p)); return DVBT_OK; }
extern "C" int dvbt_get_dims(int constellation, int hierarchy, int code_rate, int guard, int mode, dvbt_dims *o)
{
  Dims d = make_dims(constellation, hierarchy, code_rate, guard, mode);
  if (!d.valid || !o) return fail(DVBT_ERR_INVALID, "bad DVB-T parameters");
  o->fft_length = d.N; o->cp_length = d.cp; o->Kmax = d.Kmax; o->payload_length = d.payload; o->zeros_on_left = d.zl;
  o->m = d.m; o->cr_k = d.k; o->cr_n = d.n; o->norm = d.norm; o->ntraceback = d.ntb; o->info_bits_per_symbol = d.info_bits_per_symbol;
  return DVBT_OK;
}

// ------------------------------------------------------------------------------------------ helpers
template <class T> static int upload(const std::vector<T> &v, T **dptr)
{
  HIPCHK(hipMalloc((void **)dptr, v.size() * sizeof(T) + 16));
  HIPCHK(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return DVBT_OK;
}

struct DevBuf {          // growable device scratch
  void *p = nullptr; size_t cap = 0;
  int reserve(size_t n) { if (n <= cap) return DVBT_OK; if (p) (void)hipFree(p); p = nullptr; cap = 0; HIPCHK(hipMalloc(&p, n + 64)); cap = n; return DVBT_OK; }
  // grow and keep the first `keep` bytes (a block's history at the front of its input buffer).  The history was written by copies queued on
  // the call's stream `st` (the *_work_device entries never synchronise), so the move is queued on the same stream, and the old buffer is
  // released only after that stream has drained (growth happens a handful of times in a handle's life)
  int reserve_keep(size_t n, size_t keep, hipStream_t st)
  {
    if (n <= cap) return DVBT_OK;
    void *q = nullptr; HIPCHK(hipMalloc(&q, n + 64));
    if (p && keep) HIPCHK(hipMemcpyAsync(q, p, keep, hipMemcpyDeviceToDevice, st));
    if (p) { HIPCHK(hipStreamSynchronize(st)); (void)hipFree(p); }
    p = q; cap = n; return DVBT_OK;
  }
  ~DevBuf() { if (p) (void)hipFree(p); }
};

struct Tables {          // device lookup tables for one configuration
  Dims d;
  float2 *tw = nullptr; uint16_t *perm = nullptr;
  int16_t *cpilot = nullptr, *tps = nullptr; float *known = nullptr, *pref = nullptr;
  uint16_t *pay_c = nullptr, *pay_L = nullptr, *pay_R = nullptr, *tps_L = nullptr, *tps_R = nullptr;
  uint16_t *pil_k = nullptr, *pay_Li = nullptr, *pay_Ri = nullptr, *tps_Li = nullptr, *tps_Ri = nullptr; uint8_t *pay_d = nullptr, *tps_d = nullptr; int np[4] = {0, 0, 0, 0};
  uint16_t *tps_bch = nullptr;               // [7][256]: the TPS word's BCH check as a bytewise table (tps_bch_table_host)
  uint32_t *pay_pack = nullptr;              // [4][payload]: carrier | rank of the left estimation carrier << 13 | distance to it << 23 (one word per payload carrier)
  uint16_t *H = nullptr, *Hinv = nullptr; float2 *points = nullptr; uint8_t *label_tab = nullptr; int nlev = 0; float inv_step = 0.f, guard = 0.f, hshift = 0.f;
  uint8_t *mul_alpha = nullptr, *gexp = nullptr, *glog = nullptr, *prbs = nullptr;
  bool front = false, inner = false, rs = false;

  int build_fft(int N)
  {
    std::vector<float> tw_h = fft_twiddles(N);
    std::vector<float2> t2(N);
    for (int i = 0; i < N; i++) t2[i] = make_float2(tw_h[2 * i], tw_h[2 * i + 1]);
    int r = upload(t2, &tw); if (r) return r;
    return upload(fft_out_perm(N), &perm);
  }
  int build_front()
  {
    std::vector<int> c = cpilot_table(d), t = tps_table(d);
    std::vector<int16_t> c16(c.begin(), c.end()), t16(t.begin(), t.end());
    std::vector<float> pr = pilot_ref_table(d), kn(d.n_cp - 1);
    for (int i = 0; i + 1 < d.n_cp; i++) { float df = pr[c[i + 1]] - pr[c[i]]; kn[i] = df * df; }
    int r;
    if ((r = upload(c16, &cpilot)) || (r = upload(t16, &tps)) || (r = upload(pr, &pref)) || (r = upload(kn, &known))) return r;
    std::vector<uint16_t> pc, pl, prr, tl, trr, pk((size_t)4 * DEMOD_NP, 0), pli, pri, tli, tri;
    std::vector<uint8_t> pd, td;
    std::vector<uint32_t> pp;
    for (int s = 0; s < 4; s++) {
      PatternTables pt = pattern_tables(d, s);
      if ((int)pt.pay_c.size() != d.payload) return fail(DVBT_ERR_INVALID, "payload carrier table size mismatch");
      if ((int)pt.pil_k.size() > DEMOD_NP) return fail(DVBT_ERR_INVALID, "estimation carrier table overflow");
      np[s] = (int)pt.pil_k.size();
      for (size_t i = 0; i < pt.pil_k.size(); i++) pk[(size_t)s * DEMOD_NP + i] = (uint16_t)(pt.pil_k[i] | (pr[pt.pil_k[i]] < 0.f ? 0x8000 : 0));   // carrier | sign of its reference value
      pli.insert(pli.end(), pt.pay_Li.begin(), pt.pay_Li.end()); pri.insert(pri.end(), pt.pay_Ri.begin(), pt.pay_Ri.end());
      pd.insert(pd.end(), pt.pay_d.begin(), pt.pay_d.end());
      for (size_t i = 0; i < pt.pay_c.size(); i++) pp.push_back((uint32_t)pt.pay_c[i] | ((uint32_t)pt.pay_Li[i] << 13) | ((uint32_t)pt.pay_d[i] << 23));
      tli.insert(tli.end(), pt.tps_Li.begin(), pt.tps_Li.end()); tri.insert(tri.end(), pt.tps_Ri.begin(), pt.tps_Ri.end());
      td.insert(td.end(), pt.tps_d.begin(), pt.tps_d.end());
      pc.insert(pc.end(), pt.pay_c.begin(), pt.pay_c.end()); pl.insert(pl.end(), pt.pay_L.begin(), pt.pay_L.end());
      prr.insert(prr.end(), pt.pay_R.begin(), pt.pay_R.end()); tl.insert(tl.end(), pt.tps_L.begin(), pt.tps_L.end());
      trr.insert(trr.end(), pt.tps_R.begin(), pt.tps_R.end());
    }
    if ((r = upload(pc, &pay_c)) || (r = upload(pl, &pay_L)) || (r = upload(prr, &pay_R)) || (r = upload(tl, &tps_L)) || (r = upload(trr, &tps_R))) return r;
    if ((r = upload(pk, &pil_k)) || (r = upload(pli, &pay_Li)) || (r = upload(pri, &pay_Ri)) || (r = upload(pd, &pay_d)) || (r = upload(tli, &tps_Li)) ||
        (r = upload(tri, &tps_Ri)) || (r = upload(td, &tps_d)) || (r = upload(pp, &pay_pack))) return r;
    if ((r = upload(tps_bch_table_host(), &tps_bch))) return r;
    front = true;
    return DVBT_OK;
  }
  int build_inner(float gain)
  {
    int r;
    if (!H) {
      std::vector<uint16_t> h = symbol_H(d), hi(h.size());
      for (size_t q = 0; q < h.size(); q++) hi[h[q]] = (uint16_t)q;
      if ((r = upload(h, &H)) || (r = upload(hi, &Hinv))) return r;
    }
    std::vector<float> p = constellation_points(d, gain);
    std::vector<float2> p2(64, make_float2(0.f, 0.f));
    for (int i = 0; i < d.csize; i++) p2[i] = make_float2(p[2 * i], p[2 * i + 1]);
    // level grid for the candidate search: label_of[i_re * 8 + i_im].  alpha = 2, 4 (hierarchical): the same grid once the centre gap of 2 alpha
    // units is closed to 2 (every level moved (alpha - 1) units towards the centre: hshift)
    std::vector<uint8_t> lab(64, 0);
    nlev = 0; hshift = 0.f;
    if (gain > 0.f) {
      int n = 1 << (d.m / 2);
      float step = 2.0f * gain * d.norm;                       // spacing between adjacent levels
      const float sh = (float)(d.alpha - 1) * gain * d.norm;
      bool ok = true;
      for (int i = 0; i < d.csize && ok; i++) {
        const float ux = p2[i].x - (p2[i].x > 0 ? sh : -sh), uy = p2[i].y - (p2[i].y > 0 ? sh : -sh);
        float fi = ux / step + 0.5f * (n - 1), fq = uy / step + 0.5f * (n - 1);
        int ii = (int)std::lround(fi), qq = (int)std::lround(fq);
        if (ii < 0 || ii >= n || qq < 0 || qq >= n || std::fabs(fi - ii) > 1e-3f || std::fabs(fq - qq) > 1e-3f) ok = false;
        else lab[ii * 8 + qq] = (uint8_t)i;
      }
      if (ok) { nlev = n; inv_step = 1.0f / step; guard = 1.0e4f * step; hshift = sh; }
    }
    if (points) { (void)hipFree(points); points = nullptr; }
    if (label_tab) { (void)hipFree(label_tab); label_tab = nullptr; }
    if ((r = upload(p2, &points)) || (r = upload(lab, &label_tab))) return r;
    inner = true;
    return DVBT_OK;
  }
  InnerParams inner_params(int payload) const { InnerParams ip; ip.payload = payload; ip.m = d.m; ip.csize = d.csize; ip.nlev = nlev; ip.inv_step = inv_step; ip.guard = guard;
                                                  ip.hshift = hshift; ip.hier = d.hierarchy != 0 ? 1 : 0; return ip; }
  int build_rs()
  {
    std::vector<uint8_t> ex(512), lg(256), mul = rs_division_table();
    gf_tables(ex.data(), lg.data());
    int r;
    if ((r = upload(ex, &gexp)) || (r = upload(lg, &glog)) || (r = upload(mul, &mul_alpha)) || (r = upload(energy_prbs(), &prbs))) return r;
    rs = true;
    return DVBT_OK;
  }
  DemodTables demod_tables() const { DemodTables T; T.cpilot = cpilot; T.known_diff = known; T.tps = tps; T.pilot_ref = pref;
    T.pay_c = pay_c; T.pay_L = pay_L; T.pay_R = pay_R; T.tps_L = tps_L; T.tps_R = tps_R;
    T.pil_k = pil_k; for (int i = 0; i < 4; i++) T.np[i] = np[i];
    T.pay_Li = pay_Li; T.pay_Ri = pay_Ri; T.pay_d = pay_d; T.tps_Li = tps_Li; T.tps_Ri = tps_Ri; T.tps_d = tps_d; T.pay_pack = pay_pack; return T; }
  RsTables rs_tables() const { RsTables T; T.div_tab = mul_alpha; T.gexp = gexp; T.glog = glog; return T; }
  ~Tables()
  {
    void *all[] = {tw, perm, cpilot, tps, known, pref, pay_c, pay_L, pay_R, tps_L, tps_R, pil_k, pay_Li, pay_Ri, pay_d, tps_Li, tps_Ri, tps_d, pay_pack, tps_bch, H, Hinv, points, label_tab, mul_alpha, gexp, glog, prbs};
    for (void *q : all) if (q) (void)hipFree(q);
  }
};

// A7: 4 chunks per wavefront, V3_WGW independent wavefronts per workgroup (k_viterbi3.hpp)
#define V3_NTB_SWITCH(ntb, CALL) switch (ntb) { case 5: CALL(5); break; case 9: CALL(9); break; case 10: CALL(10); break; case 15: CALL(15); break; default: CALL(24); break; }
// device buffers of the proof and the repair passes; one launch of the decoder at a time per owner (stream order)
struct VitProof {
  int *snap = nullptr, *ctl = nullptr; long long cap = 0;
  int reserve(long long chunks)
  {
    if (chunks <= cap) return DVBT_OK;
    if (snap) (void)hipFree(snap); if (ctl) (void)hipFree(ctl);
    snap = ctl = nullptr; cap = 0;
    HIPCHK(hipMalloc((void **)&snap, v3_snap_words(chunks) * sizeof(int))); HIPCHK(hipMalloc((void **)&ctl, v3_ctl_words(chunks) * sizeof(int)));
    HIPCHK(hipMemset(ctl, 0, v3_ctl_words(chunks) * sizeof(int)));
    cap = chunks; return DVBT_OK;
  }
  V3Aux aux() const { V3Aux a; memset(&a, 0, sizeof a); a.snap = snap; a.ctl = ctl; a.cap = cap; a.carry_rel = -1; return a; }
  ~VitProof() { if (snap) (void)hipFree(snap); if (ctl) (void)hipFree(ctl); }
};
enum { VIT_PLAIN = -1, VIT_REPAIR = 0, VIT_REPAIR_COUNT = 1, VIT_PROOF_ONLY = 2, VIT_FORCE_SEQ = 3, VIT_FORCE_CONFLICT = 4 };   // dvbt_rx_params.viterbi_verify
constexpr unsigned V3_REPAIR_GRID = 64;   // workgroups of the parallel repair pass: 1024 decoders at once, the rest in rounds (a wavefront without a listed chunk returns at once)

// ax == null or mode == VIT_PLAIN: the chunk decoders alone.  Otherwise the launch that is the streaming decoder by construction: the decoders leave their states, the checker
// lists the chunks that are not proven, the repair passes decode those again (k_viterbi3.hpp); nothing is read back.
static void launch_viterbi(hipStream_t s, const uint8_t *in, uint8_t *out, const RxState *st, long long steps_fixed, const VitParams &vp,
                           long long in_base, long long out_lo, long long max_out_bytes, const V3Aux *ax = nullptr, int mode = VIT_PLAIN)
{
  const bool proof = ax && mode != VIT_PLAIN;
  const long long grid0 = proof ? ax->grid0 : out_lo;
  long long chunks = (out_lo + max_out_bytes - grid0 + vp.chunk_bytes - 1) / vp.chunk_bytes;
  if (chunks < 1) chunks = 1;
  const dim3 grid((unsigned)((chunks + 4 * V3_WGW - 1) / (4 * V3_WGW))), blk(64 * V3_WGW);
  V3Aux none; memset(&none, 0, sizeof none);
  if (proof) {
    V3Aux a = *ax;
    if (mode == VIT_FORCE_CONFLICT) { a.debug = 1; mode = VIT_REPAIR_COUNT; }
    const dim3 cgrid((unsigned)((chunks + 255) / 256));
    if (vp.warm == V3_WARM) {
#define V3_CALL(N) hipLaunchKernelGGL((viterbi3_kernel<N, V3_WARM, 1>), grid, blk, 0, s, in, out, st, steps_fixed, vp, a, in_base, out_lo)
      V3_NTB_SWITCH(vp.ntb, V3_CALL)
#undef V3_CALL
    } else {
#define V3_CALL(N) hipLaunchKernelGGL((viterbi3_kernel<N, 0, 1>), grid, blk, 0, s, in, out, st, steps_fixed, vp, a, in_base, out_lo)
      V3_NTB_SWITCH(vp.ntb, V3_CALL)
#undef V3_CALL
    }
    if (chunks <= V3_FIX_SMALL && mode != VIT_PROOF_ONLY) {   // few chunks: the three passes as ONE launch of one workgroup
#define V3_CALL(N) hipLaunchKernelGGL(viterbi_fix_kernel<N>, dim3(1), blk, 0, s, in, out, st, steps_fixed, vp, a, in_base, out_lo, mode == VIT_FORCE_SEQ ? 1 : 0)
      V3_NTB_SWITCH(vp.ntb, V3_CALL)
#undef V3_CALL
      if (mode != VIT_REPAIR) {
        hipLaunchKernelGGL(viterbi_check_kernel, dim3(1), dim3(256), 0, s, a, st, steps_fixed, vp, 1);
        hipLaunchKernelGGL(viterbi_count_kernel, cgrid, dim3(256), 0, s, a, st, steps_fixed, vp);
      }
      return;
    }
    hipLaunchKernelGGL(viterbi_check_kernel, cgrid, dim3(256), 0, s, a, st, steps_fixed, vp, 0);
    if (mode == VIT_REPAIR || mode == VIT_REPAIR_COUNT) {
#define V3_CALL(N) hipLaunchKernelGGL(viterbi_repair_kernel<N>, dim3(V3_REPAIR_GRID), blk, 0, s, in, out, st, steps_fixed, vp, a, in_base, out_lo)
      V3_NTB_SWITCH(vp.ntb, V3_CALL)
#undef V3_CALL
    }
    if (mode != VIT_PROOF_ONLY) {
      const int force = mode == VIT_FORCE_SEQ ? 1 : 0;
#define V3_CALL(N) hipLaunchKernelGGL(viterbi_repair_seq_kernel<N>, dim3(1), dim3(64), 0, s, in, out, st, steps_fixed, vp, a, in_base, out_lo, force)
      V3_NTB_SWITCH(vp.ntb, V3_CALL)
#undef V3_CALL
    }
    if (mode != VIT_REPAIR) {   // the final check: chunks that are still not proven (none behind the repair passes)
      hipLaunchKernelGGL(viterbi_check_kernel, dim3(1), dim3(256), 0, s, a, st, steps_fixed, vp, 1);
      hipLaunchKernelGGL(viterbi_count_kernel, cgrid, dim3(256), 0, s, a, st, steps_fixed, vp);
    }
    return;
  }
  if (vp.warm != V3_WARM) {   // a warm-up other than the default (dvbt_rx_params.viterbi_warm_windows): the instantiation that reads it from the parameters
#define V3_CALL(N) hipLaunchKernelGGL((viterbi3_kernel<N, 0, 0>), grid, blk, 0, s, in, out, st, steps_fixed, vp, none, in_base, out_lo)
    V3_NTB_SWITCH(vp.ntb, V3_CALL)
#undef V3_CALL
    return;
  }
#define V3_CALL(N) hipLaunchKernelGGL((viterbi3_kernel<N, V3_WARM, 0>), grid, blk, 0, s, in, out, st, steps_fixed, vp, none, in_base, out_lo)
  V3_NTB_SWITCH(vp.ntb, V3_CALL)   // the traceback depth is a template parameter (hop schedule fixed at compile time)
#undef V3_CALL
}

// the plan of A8 + A9 + descrambler for a Viterbi stream that the host has laid out from several lock periods (segment_periods)
__global__ void tail_patch_kernel(RxState *st, long long items)
{
  st->n_rs_items = items; st->n_rs_words = items * 8; st->stream_rs_items = items;
  st->rs_fail = 0; st->rs_corr = 0; st->rs_list_n = 0; st->sym_off = 0;
}

static FrontParams make_front_params(const Dims &d, float snr_db)
{
  FrontParams p; memset(&p, 0, sizeof p);
  p.N = d.N; p.cp = d.cp; p.K = d.K; p.zl = d.zl; p.payload = d.payload; p.n_cp = d.n_cp; p.n_tps = d.n_tps; p.fi_start = d.fi_start;
  p.R = ACQ_R;
  float snr = (float)pow(10, snr_db / 10.0);               // ofdm_sym_acquisition_impl.cc:390-391
  float rho = (float)(snr / (snr + 1.0));
  p.half_rho = (float)(rho / 2.0);
  return p;
}
static int set_lds(const void *fn, size_t bytes)
{
  if (bytes > 48 * 1024) HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return DVBT_OK;
}

// ------------------------------------------------------------------------------------------ segment API
enum { ST_ACQ = 0, ST_FFT, ST_DEMOD, ST_INNER, ST_VIT, ST_RS, ST_END, ST_COUNT };
static const char *kStageNames[] = {"acq", "fft", "demod", "inner", "viterbi", "rs"};

// taps and polyphase branches of the stock rational_resampler (dvbt_tables.hpp::resampler_taps) on the device
struct ResamplerDesign {
  int ri = 0, rd = 0, nt = 0; std::vector<float> taps; float *br = nullptr;
  int build(int interp, int decim)
  {
    if (interp <= 0 || decim <= 0) return fail(DVBT_ERR_INVALID, "bad resampler ratio");
    taps = resampler_taps(interp, decim, ri, rd);
    std::vector<float> b = resampler_branches(taps, ri, nt);
    if ((long long)ri * nt > RS_MAX_BRANCH_FLOATS || 256ll * rd / ri + nt + 2 > RS_TILE_IN) return fail(DVBT_ERR_INVALID, "resampler ratio outside the supported range");
    HIPCHK(hipMalloc((void **)&br, b.size() * sizeof(float)));
    HIPCHK(hipMemcpy(br, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice));
    return DVBT_OK;
  }
  ~ResamplerDesign() { if (br) (void)hipFree(br); }
};

struct dvbt_rx {
  dvbt_rx_params prm; Dims d; Tables T; FrontParams fp; VitParams vp;
  hipStream_t own_stream = nullptr, cur_stream = nullptr;
  hipStream_t front_stream = nullptr; hipEvent_t front_ev[2] = {nullptr, nullptr};   // dvbt_rx_params.front_priority: the front end's high-priority stream, fork / join
  size_t max_samples = 0; int max_calls = 0;
  float2 *d_iq = nullptr;              // only when input comes from the host
  ResamplerDesign rsd; float2 *rs_iq = nullptr; size_t chain_max = 0;
  AcqState *acq_carry = nullptr; bool use_carry = false;     // peak-detector state carried into a restart after a lost lock   // front-of-chain resample + scale (next row 2): its output buffer
  float2 *g_init = nullptr; float *l_init = nullptr; float2 *g_trk = nullptr; float *l_trk = nullptr;
  SymMeta *meta = nullptr; RxState *st = nullptr, *st_host = nullptr; TpsState *tps_state = nullptr;
  // two acquisition contexts (state block + per-symbol tracker results): `st` / `meta` point at the one in use.  The lock-period walk searches on in the
  // other one, so that what the last decoded period left behind stays valid for the byte de-interleaver, the report and the taps (segment_periods)
  static constexpr int NCTX = 3;            // (segment_periods uses two; the streaming entry's walk keeps a found period, decodes the one before it and searches for the next one at once)
  RxState *st_ctx[NCTX] = {nullptr, nullptr, nullptr}; SymMeta *meta_ctx[NCTX] = {nullptr, nullptr, nullptr}; int ctx = 0;
  // ... and what else an acquisition of one period and the decode of the period before it would share if they ran at the same time -- which they do (segment_periods:
  // the search for period p + 1 on the caller's stream, the decode of period p on aux_stream): the trackers' flag words, the symbol kernel's ticket, the drift
  // model's verdict, the page-locked copy of the state block.  trk_flags / sym_ticket / drift.flags / st_host point at the context in use (set_ctx)
  int *trk_flags_ctx[NCTX] = {nullptr, nullptr, nullptr}, *sym_ticket_ctx[NCTX] = {nullptr, nullptr, nullptr}, *drift_flags_ctx[NCTX] = {nullptr, nullptr, nullptr}; RxState *st_host_ctx[NCTX] = {nullptr, nullptr, nullptr};
  hipStream_t aux_stream = nullptr; hipEvent_t aux_ev = nullptr;
  int *trk_cp_a = nullptr, *trk_cp_b = nullptr, *trk_flags = nullptr; float *trk_eps = nullptr; TpsEdge *tps_edges = nullptr;
  int *centre = nullptr, *anchor_pos = nullptr;   // predicted CP position per call / coarse estimates every ACQ_ANCHOR calls
  float2 *acq_tap = nullptr, *fft_out = nullptr, *eq = nullptr, *tpsval = nullptr; SymInfo *info = nullptr; int *maj = nullptr, *sym_index = nullptr;
  uint8_t *labels = nullptr, *symdeint_tap = nullptr, *bitdeint = nullptr, *vit = nullptr, *deint_tap = nullptr, *rs_out = nullptr, *ts_out = nullptr;
  uint8_t *bitdeint_lp = nullptr;           // hierarchical modes: the bit de-interleaver's second output
  uint8_t *bd_log = nullptr; std::vector<dvbt_period_tap> plog; size_t plog_bytes = 0;   // dvbt_rx_enable_taps(h, 2): every lock period's decoder input
  VitProof vproof; bool vit_checked = false;   // the Viterbi stage's proof + repair passes (dvbt_rx_params.viterbi_verify): the chunk decoders' states, the passes' counters
  size_t vit_cap = 0; RsDefer *rs_defer = nullptr; int rs_defer_cap = 0;
  unsigned long long *rs_sync = nullptr;     // bit w: payload byte 0 of RS word w is 0xB8 (deint_rs_kernel / rs_fix_kernel -> descramble_scan_kernel)
  unsigned soft_grid = 0;     // workgroups the decision scratch has slots for
  float *csi = nullptr; int8_t *soft_a = nullptr; uint16_t *soft_tab = nullptr; unsigned *soft_scratch = nullptr;   // soft-decision mode (k_soft.hpp): channel state per carrier, soft values, A5 + A6 gather table, decision slots
  int timing = 0;             // 0 off, 1 events around every stage, 2 around the decoder only (dvbt_rx_enable_timing)
  bool pending = false;
  hipEvent_t ev[ST_COUNT]; double acc_ms[ST_COUNT] = {0}; long n_timed = 0; bool ev_ready = false, ev_recorded = false;
  dvbt_rx_report last; bool have_last = false;
  dvbt_rx_cut cut = {0, 0, 0};
  float2 *tps_prev = nullptr, *tps_prev_snap[2] = {nullptr, nullptr}; TpsState *tps_snap[2] = {nullptr, nullptr};  DescrRun *descr_runs = nullptr; int *descr_nruns = nullptr;
  int n_periods = 1; size_t seg_offset = 0;
  std::vector<dvbt_lock_period> periods;    // phase A of the last synchronous run
  DriftBufs drift = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; double *drift_mem = nullptr;   // k_drift.hpp
  struct GraphEntry { const void *iq; size_t n; hipStream_t s; long long sym_off; int delay, phase; hipGraphExec_t exec; };
  std::vector<GraphEntry> graphs;           // launch_graph: the captured launch sequences (a handful: a receiver's ring of segments)
  int ncu = 256;
  long long n_small = 0, n_general = 0;     // acquisition-only passes of the lock-period walk through acq_small_kernel / through the general kernels (dvbt_rx_walk_stats)
  int sym_grid = 512;                       // workgroups of symbol8k_kernel (two per CU)
  int *sym_ticket = nullptr;                // its symbol counter
};

constexpr long long kVitMaxChunk = 3000;
static int vit_chunk_bytes(const dvbt_rx *h, long long max_vit);

// launch_graph: a captured launch sequence holds the handle's buffers and choices as they were when it was captured -- the tap buffers and the `taps` instantiations, the
// acquisition context in use (state block, tracker flags, ticket, page-locked copy), the stage events.  Whatever changes one of them drops the graphs.
static void drop_graphs(dvbt_rx *h)
{
  for (auto &ge : h->graphs) if (ge.exec) (void)hipGraphExecDestroy(ge.exec);
  h->graphs.clear();
}

static void rx_free(dvbt_rx *h)
{
  void *all[] = {h->bd_log, h->bitdeint_lp, h->st_ctx[0], h->st_ctx[1], h->st_ctx[2], h->meta_ctx[0], h->meta_ctx[1], h->meta_ctx[2], h->tps_prev_snap[0], h->tps_prev_snap[1], h->tps_snap[0], h->tps_snap[1], h->csi, h->soft_a, h->soft_tab, h->soft_scratch, h->rs_defer, h->rs_sync, h->drift_mem, h->drift.delta, h->drift_flags_ctx[0], h->drift_flags_ctx[1], h->drift_flags_ctx[2], h->tps_prev, h->descr_runs, h->descr_nruns, h->centre, h->anchor_pos, h->sym_ticket_ctx[0], h->sym_ticket_ctx[1], h->sym_ticket_ctx[2], h->tps_edges, h->trk_cp_a, h->trk_cp_b, h->trk_flags_ctx[0], h->trk_flags_ctx[1], h->trk_flags_ctx[2], h->trk_eps, h->d_iq, h->rs_iq, h->acq_carry, h->g_init, h->l_init, h->g_trk, h->l_trk, h->tps_state, h->acq_tap, h->fft_out, h->eq, h->tpsval,
                 h->info, h->maj, h->sym_index, h->labels, h->symdeint_tap, h->bitdeint, h->vit, h->deint_tap, h->rs_out, h->ts_out};
  for (void *q : all) if (q) (void)hipFree(q);
  for (auto &ge : h->graphs) if (ge.exec) (void)hipGraphExecDestroy(ge.exec);
  for (int i = 0; i < dvbt_rx::NCTX; i++) if (h->st_host_ctx[i]) (void)hipHostFree(h->st_host_ctx[i]);
  if (h->aux_ev) (void)hipEventDestroy(h->aux_ev);
  if (h->aux_stream) (void)hipStreamDestroy(h->aux_stream);
  if (h->ev_ready) for (int i = 0; i < ST_COUNT; i++) (void)hipEventDestroy(h->ev[i]);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  for (int i = 0; i < 2; i++) if (h->front_ev[i]) (void)hipEventDestroy(h->front_ev[i]);
  if (h->front_stream) (void)hipStreamDestroy(h->front_stream);
  delete h;
}

extern "C" int dvbt_rx_create(const dvbt_rx_params *p, dvbt_rx **out)
{
  if (!p || !out) return fail(DVBT_ERR_INVALID, "null argument");
  int r = need_device(); if (r) return r;
  Dims d = make_dims(p->constellation, p->hierarchy, p->code_rate, p->guard_interval, p->transmission_mode);
  if (!d.valid) return fail(DVBT_ERR_INVALID, "bad DVB-T parameters");
  if (p->viterbi_bsize <= 0 || (2 * d.k * p->viterbi_bsize) % 16 != 0 || (p->viterbi_bsize * d.n) % d.m != 0)
    return fail(DVBT_ERR_INVALID, "viterbi_bsize must make bsize*n/m integral and 2*k*bsize a multiple of 16");
  if (d.hierarchy != 0 && d.m == 2) return fail(DVBT_ERR_INVALID, "hierarchical QPSK does not exist (the reference's bit_inner_deinterleaver divides by d_v - 2 = 0 in its constructor)");
  if (d.hierarchy != 0 && p->soft_decision) return fail(DVBT_ERR_INVALID, "soft decisions are built for the non-hierarchical modes");
  if (p->hier_stream < 0 || p->hier_stream > 1) return fail(DVBT_ERR_INVALID, "hier_stream must be 0 (HP) or 1 (LP)");
  if (p->viterbi_warm_windows != 0 && (p->viterbi_warm_windows < 2 * V3_BLK || p->viterbi_warm_windows > V3_WARM_MAX || p->viterbi_warm_windows % V3_BLK != 0))
    return fail(DVBT_ERR_INVALID, "viterbi_warm_windows must be 0 (default) or a multiple of 24 in [48, 1152]");
  if (p->viterbi_warm_windows != 0 && p->soft_decision) return fail(DVBT_ERR_INVALID, "viterbi_warm_windows applies to the hard-decision decoder");
  if (p->viterbi_verify < VIT_PLAIN || p->viterbi_verify > VIT_FORCE_CONFLICT) return fail(DVBT_ERR_INVALID, "viterbi_verify must lie in [-1, 4]");
  if (p->viterbi_verify > 0 && p->soft_decision) return fail(DVBT_ERR_INVALID, "viterbi_verify applies to the hard-decision decoder");
  HIPCHK(hipSetDevice(p->device));
  dvbt_rx *h = new dvbt_rx();
  h->prm = *p; h->d = d; h->T.d = d;
  // capacity of the chain proper, in samples at the OFDM elementary rate
  size_t chain_max = p->max_samples;
  if (p->resample_interp > 0 || p->resample_decim > 0) {
    int r2 = h->rsd.build(p->resample_interp, p->resample_decim); if (r2) { delete h; return r2; }
    chain_max = (size_t)(((unsigned long long)p->max_samples * h->rsd.ri + h->rsd.rd - 1) / h->rsd.rd);
  }
  if (chain_max < (size_t)(2 * d.N + d.cp + 16)) { delete h; return fail(DVBT_ERR_INVALID, "max_samples smaller than one acquisition window"); }
  h->chain_max = chain_max;
  h->fp = make_front_params(d, p->snr_db);
  int cb = p->viterbi_chunk_bytes > 0 ? p->viterbi_chunk_bytes : 768;
  h->vp = make_vit_params(d, p->viterbi_bsize, cb);
  if (p->viterbi_warm_windows > 0) h->vp.warm = p->viterbi_warm_windows;
  h->max_samples = p->max_samples;
  h->max_calls = (int)((chain_max - (2 * d.N + d.cp + 16)) / (d.N + d.cp) + 1);
#define RXCHK(x) do { int r_ = (x); if (r_) { rx_free(h); return r_; } } while (0)
#define RXHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_err = std::string(#x) + ": " + hipGetErrorString(e_); rx_free(h); return DVBT_ERR_HIP; } } while (0)
  RXCHK(h->T.build_fft(d.N)); RXCHK(h->T.build_front()); RXCHK(h->T.build_inner(1.0f)); RXCHK(h->T.build_rs());
  RXHIP(hipStreamCreate(&h->own_stream));
  if (p->front_priority) {
    int least = 0, greatest = 0; RXHIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    RXHIP(hipStreamCreateWithPriority(&h->front_stream, hipStreamNonBlocking, greatest));
    for (int i = 0; i < 2; i++) RXHIP(hipEventCreateWithFlags(&h->front_ev[i], hipEventDisableTiming));
  }
  if (h->rsd.ri) RXHIP(hipMalloc((void **)&h->rs_iq, sizeof(float2) * (chain_max + 16)));
  const size_t C = (size_t)h->max_calls, N = d.N, P = d.payload;
  RXHIP(hipMalloc((void **)&h->g_init, sizeof(float2) * ACQ_INIT_TRIES_MAX * N)); RXHIP(hipMalloc((void **)&h->l_init, sizeof(float) * ACQ_INIT_TRIES_MAX * N));
  RXHIP(hipMalloc((void **)&h->g_trk, sizeof(float2) * C * 2 * ACQ_R)); RXHIP(hipMalloc((void **)&h->l_trk, sizeof(float) * C * 2 * ACQ_R));
  for (int i = 0; i < dvbt_rx::NCTX; i++) { RXHIP(hipMalloc((void **)&h->meta_ctx[i], sizeof(SymMeta) * C)); RXHIP(hipMalloc((void **)&h->st_ctx[i], sizeof(RxState))); RXHIP(hipMemset(h->st_ctx[i], 0, sizeof(RxState))); }
  h->meta = h->meta_ctx[0]; h->st = h->st_ctx[0]; h->ctx = 0;
  RXHIP(hipMalloc((void **)&h->trk_cp_a, sizeof(int) * C)); RXHIP(hipMalloc((void **)&h->trk_cp_b, sizeof(int) * C));
  RXHIP(hipMalloc((void **)&h->trk_eps, sizeof(float) * C));
  for (int i = 0; i < dvbt_rx::NCTX; i++) { RXHIP(hipMalloc((void **)&h->trk_flags_ctx[i], sizeof(int) * 16)); RXHIP(hipMemset(h->trk_flags_ctx[i], 0, sizeof(int) * 16)); }
  h->trk_flags = h->trk_flags_ctx[0];
  RXHIP(hipStreamCreate(&h->aux_stream)); RXHIP(hipEventCreateWithFlags(&h->aux_ev, hipEventDisableTiming));
  RXHIP(hipMalloc((void **)&h->tps_prev, sizeof(float2) * d.n_tps)); RXHIP(hipMemset(h->tps_prev, 0, sizeof(float2) * d.n_tps));
  for (int i = 0; i < 2; i++) { RXHIP(hipMalloc((void **)&h->tps_prev_snap[i], sizeof(float2) * d.n_tps)); RXHIP(hipMalloc((void **)&h->tps_snap[i], sizeof(TpsState))); }
  RXHIP(hipMalloc((void **)&h->descr_runs, sizeof(DescrRun) * DESCR_MAX_RUNS)); RXHIP(hipMalloc((void **)&h->descr_nruns, sizeof(int)));
  RXHIP(hipMalloc((void **)&h->centre, sizeof(int) * (C + 1))); RXHIP(hipMalloc((void **)&h->anchor_pos, sizeof(int) * (C / ACQ_ANCHOR + 4)));
  RXHIP(hipMalloc((void **)&h->tps_edges, sizeof(TpsEdge) * (C / TPS_SEG + 2)));
  for (int i = 0; i < dvbt_rx::NCTX; i++) { RXHIP(hipHostMalloc((void **)&h->st_host_ctx[i], sizeof(RxState))); memset(h->st_host_ctx[i], 0, sizeof(RxState)); }
  h->st_host = h->st_host_ctx[0];
  RXHIP(hipMalloc((void **)&h->acq_carry, sizeof(AcqState))); RXHIP(hipMalloc((void **)&h->tps_state, sizeof(TpsState)));
  RXHIP(hipMalloc((void **)&h->labels, C * P + 64));   // A1..A4 are one kernel: a symbol reaches HBM as label bytes; fft_out and eq exist only as debug taps
  RXHIP(hipMalloc((void **)&h->tpsval, sizeof(float2) * C * d.n_tps)); RXHIP(hipMalloc((void **)&h->info, sizeof(SymInfo) * C));
  RXHIP(hipMalloc((void **)&h->maj, sizeof(int) * C)); RXHIP(hipMalloc((void **)&h->sym_index, sizeof(int) * C));
  RXHIP(hipMalloc((void **)&h->bitdeint, C * P + 64));
  if (d.hierarchy != 0) RXHIP(hipMalloc((void **)&h->bitdeint_lp, C * P + 64));
  { int ncu = 256; (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->prm.device); h->sym_grid = s8_grid(ncu); h->ncu = ncu; }
  if (p->viterbi_verify != VIT_PLAIN && !p->soft_decision) {   // the proof's state slots for the most chunks a launch of this handle can have (vit_chunk_bytes)
    const long long mv = (long long)C * P * d.m * d.k / (8 * d.n) + 1;
    long long chunks;
    if (p->viterbi_chunk_bytes > 0) chunks = mv / vit_chunk_bytes(h, mv) + 2;
    else { const long long slots = (long long)h->ncu * V3_WAVES_PER_CU * 4; chunks = slots * ((mv + slots * kVitMaxChunk - 1) / (slots * kVitMaxChunk)) + 4; }
    RXCHK(h->vproof.reserve(chunks));
  }
  h->vit_cap = C * P * d.m * d.k / (8 * d.n) + 4096 + (2u << 20);   // (+ 2 MB: a walk of the streaming entry carries the tail of its Viterbi stream from window to window)
  h->rs_defer_cap = (int)((h->vit_cap / 204 / 64 + 2) * (RS_LANE_MIN - 1));
  RXHIP(hipMalloc((void **)&h->rs_defer, sizeof(RsDefer) * (size_t)h->rs_defer_cap));
  RXHIP(hipMalloc((void **)&h->rs_sync, sizeof(unsigned long long) * (h->vit_cap / 204 / 64 + 2)));
  RXHIP(hipMalloc((void **)&h->vit, h->vit_cap)); RXHIP(hipMalloc((void **)&h->rs_out, h->vit_cap)); RXHIP(hipMalloc((void **)&h->ts_out, h->vit_cap));
  RXHIP(hipMemset(h->st, 0, sizeof(RxState)));
  for (int i = 0; i < ST_COUNT; i++) RXHIP(hipEventCreate(&h->ev[i]));
  h->ev_ready = true;
  for (int i = 0; i < dvbt_rx::NCTX; i++) { RXHIP(hipMalloc((void **)&h->sym_ticket_ctx[i], 64)); RXHIP(hipMemset(h->sym_ticket_ctx[i], 0, 64)); }
  h->sym_ticket = h->sym_ticket_ctx[0];
  if (p->soft_decision) {
    RXHIP(hipMalloc((void **)&h->eq, sizeof(float2) * C * P)); RXHIP(hipMalloc((void **)&h->csi, sizeof(float) * C * P));
    RXHIP(hipMalloc((void **)&h->soft_a, C * P * d.m + 64)); RXHIP(hipMalloc((void **)&h->soft_tab, sizeof(uint16_t) * 2 * P * d.m));
    // decision slots: one per resident wavefront of the largest launch this handle can make (max_samples), not of the largest launch there is
    // (2048 workgroups = 704 MB: what a 17-superframe segment of 8k QAM64 7/8 uses; a 2-superframe handle of the streaming entry needs 1/8 of that)
    const long long max_vit_soft = (long long)C * P * d.m * d.k / (8 * d.n) + 1;
    h->soft_grid = s4_grid_bound(max_vit_soft);               // (s4_grid itself is not monotone in the stream's length: ADVICE r04)
    RXHIP(hipMalloc((void **)&h->soft_scratch, sizeof(unsigned) * (size_t)h->soft_grid * S4_WAVES * S4_SLOT_WORDS));
    hipLaunchKernelGGL(soft_tab_kernel, dim3(64), dim3(256), 0, h->own_stream, h->T.inner_params(d.payload), (const uint16_t *)h->T.H, (const uint16_t *)h->T.Hinv, h->soft_tab);
    RXHIP(hipStreamSynchronize(h->own_stream));
  }
  {   // the float phase accumulator's wander (k_drift.hpp): tables and sums per call, one deviation per 32 samples of every item
    RXHIP(hipMalloc((void **)&h->drift_mem, sizeof(double) * (C * (DRIFT_TAB + 4) + 8)));
    h->drift.tabs = h->drift_mem; h->drift.ex_run = h->drift.tabs + C * DRIFT_TAB; h->drift.ex_entry = h->drift.ex_run + C;
    h->drift.d = h->drift.ex_entry + C; h->drift.S = h->drift.d + C; h->drift.A0 = h->drift.S + C;
    RXHIP(hipMalloc((void **)&h->drift.delta, sizeof(float) * C * (N / 32))); for (int i = 0; i < dvbt_rx::NCTX; i++) { RXHIP(hipMalloc((void **)&h->drift_flags_ctx[i], 16)); RXHIP(hipMemset(h->drift_flags_ctx[i], 0, 16)); }
    h->drift.flags = h->drift_flags_ctx[0];
  }
  RXCHK(set_lds((const void *)symbol8k_kernel<false, false>, S8_LDS_BYTES)); RXCHK(set_lds((const void *)symbol8k_kernel<false, true>, S8_LDS_BYTES));
  RXCHK(set_lds((const void *)symbol8k_kernel<true, false>, S8_LDS_BYTES)); RXCHK(set_lds((const void *)symbol8k_kernel<true, true>, S8_LDS_BYTES));
  RXCHK(set_lds((const void *)symbol2k_kernel<false, false>, S2_LDS_BYTES)); RXCHK(set_lds((const void *)symbol2k_kernel<false, true>, S2_LDS_BYTES));
  RXCHK(set_lds((const void *)symbol2k_kernel<true, false>, S2_LDS_BYTES)); RXCHK(set_lds((const void *)symbol2k_kernel<true, true>, S2_LDS_BYTES));
  RXCHK(set_lds((const void *)inner_kernel<6>, inner_lds_bytes(P)));
  RXCHK(set_lds((const void *)acq_anchor_kernel, acq_anchor_lds_bytes((int)N, d.cp)));
  RXCHK(set_lds((const void *)acq_small_kernel, (size_t)acq_small_cpc(d.cp) * 2 * (d.cp + 2 * ACQ_R) * sizeof(float2)));
  *out = h;
  return DVBT_OK;
}

extern "C" int dvbt_rx_set_cut(dvbt_rx *h, const dvbt_rx_cut *cut)
{
  if (!h || !cut) return fail(DVBT_ERR_INVALID, "null argument");
  if (cut->stream_symbol_offset < 0 || cut->stream_symbol_offset % 272) return fail(DVBT_ERR_INVALID, "stream_symbol_offset must be a non-negative multiple of 272 (whole superframes)");
  if (cut->start_delay_symbols < 0 || cut->start_delay_symbols >= 272 || cut->descr_call_phase < 0 || cut->descr_call_phase > 16) return fail(DVBT_ERR_INVALID, "start_delay_symbols must lie in [0, 272), descr_call_phase in [0, 16]");
  h->cut = *cut;
  return DVBT_OK;
}

extern "C" int dvbt_rx_enable_timing(dvbt_rx *h, int enable)
{
  if (!h) return DVBT_ERR_INVALID;
  h->timing = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
  drop_graphs(h);
  if (enable) { for (int i = 0; i <= ST_END; i++) h->acc_ms[i] = 0.0; h->n_timed = 0; }     // a new measurement window: dvbt_rx_stage_ms averages from here on
  return DVBT_OK;
}

// The buffers behind the Viterbi decoder for a stream of at least `cap` bytes (contents kept).  A handle is created for one launch over max_samples; the streaming entry's
// walk lays the lock periods of a WINDOW out in h->vit -- up to two pieces and a threshold of samples when a lost piece is harvested late -- in front of an open period that
// alone may fill a launch (dvbt_rx_stream_push failed with DVBT_ERR_CAPACITY there).  Grows, never shrinks; synchronises the device (a walk is synchronous anyway).
static int rx_reserve_vit(dvbt_rx *h, size_t cap)
{
  if (cap <= h->vit_cap) return DVBT_OK;
  HIPCHK(hipDeviceSynchronize());
  drop_graphs(h);
  const size_t old = h->vit_cap;
  uint8_t **bufs[] = {&h->vit, &h->rs_out, &h->ts_out, &h->deint_tap};
  for (uint8_t **b : bufs) {
    if (!*b) continue;                                             // (deint_tap: a debug tap, there only on request)
    uint8_t *q = nullptr; HIPCHK(hipMalloc((void **)&q, cap));
    HIPCHK(hipMemcpy(q, *b, old, hipMemcpyDeviceToDevice));
    (void)hipFree(*b); *b = q;
  }
  const int dcap = (int)((cap / 204 / 64 + 2) * (RS_LANE_MIN - 1));
  { RsDefer *q = nullptr; HIPCHK(hipMalloc((void **)&q, sizeof(RsDefer) * (size_t)dcap)); (void)hipFree(h->rs_defer); h->rs_defer = q; h->rs_defer_cap = dcap; }
  { unsigned long long *q = nullptr; HIPCHK(hipMalloc((void **)&q, sizeof(unsigned long long) * (cap / 204 / 64 + 2)));
    HIPCHK(hipMemcpy(q, h->rs_sync, sizeof(unsigned long long) * (old / 204 / 64 + 2), hipMemcpyDeviceToDevice)); (void)hipFree(h->rs_sync); h->rs_sync = q; }
  h->vit_cap = cap;
  return DVBT_OK;
}

// debug taps that are not pipeline buffers are allocated on first request
static int ensure_taps(dvbt_rx *h)
{
  const size_t C = (size_t)h->max_calls, N = h->d.N, P = h->d.payload;
  if (!h->acq_tap) HIPCHK(hipMalloc((void **)&h->acq_tap, sizeof(float2) * C * N));
  if (!h->fft_out) HIPCHK(hipMalloc((void **)&h->fft_out, sizeof(float2) * C * N));
  if (!h->eq) HIPCHK(hipMalloc((void **)&h->eq, sizeof(float2) * C * P));
  if (!h->symdeint_tap) HIPCHK(hipMalloc((void **)&h->symdeint_tap, C * P + 64));
  if (!h->deint_tap) HIPCHK(hipMalloc((void **)&h->deint_tap, h->vit_cap));
  return DVBT_OK;
}
extern "C" int dvbt_rx_enable_taps(dvbt_rx *h, int enable)
{
  if (!h) return DVBT_ERR_INVALID;
  if (h->pending) return fail(DVBT_ERR_STATE, "dvbt_rx_enable_taps: a segment is in flight (dvbt_rx_segment_finish first)");
  drop_graphs(h);
  if (enable == 2 && !h->bd_log) HIPCHK(hipMalloc((void **)&h->bd_log, (size_t)h->max_calls * h->d.payload + 64));
  if (enable) return ensure_taps(h);
  if (h->bd_log) { (void)hipFree(h->bd_log); h->bd_log = nullptr; }
  // in soft-decision mode eq is not a debug tap but the soft demapper's input (allocated at create; it also selects the symbol kernel's instantiation that writes eq and csi)
  void **all[] = {(void **)&h->acq_tap, (void **)&h->fft_out, h->prm.soft_decision ? (void **)nullptr : (void **)&h->eq, (void **)&h->symdeint_tap, (void **)&h->deint_tap};
  for (void **q : all) if (q && *q) { (void)hipFree(*q); *q = nullptr; }
  return DVBT_OK;
}

// How one lock period of a segment is processed.  The asynchronous entry (dvbt_rx_segment_enqueue_device) runs one period with the
// defaults; the synchronous entries walk the segment's lock periods (segment_periods).
struct EnqOpt {
  bool acq_only = false;      // ofdm_sym_acquisition alone (where does the lock start, how long does it hold)
  int init_tries = ACQ_INIT_TRIES;   // windows the initial search examines before it gives up (the reference consumes them one by one)
  bool skip_acq = false;      // the acquisition results of the acq_only run just before (same iq, same hist, same carry) are still in the handle: go on from there
  bool use_carry = false;     // the peak detector's average is carried in from the call that lost the previous lock ...
  float carry_avg = 0.f;      // ... this value
  long long hist = 0;         // samples of the stream in memory in front of iq[0]
  bool continuation = false;  // not the first period that reaches demod_reference_signals: its TPS state and the previous symbol's TPS carriers are
                              // carried over, the first item bears the sync_start tag (demod_reference_signals_impl.cc:115-116)
  bool keep_last = false;     // a later period delivers items: the last item of this one leaves the demodulator too
  size_t vit_off = 0;         // where this period's decoded bytes go in the Viterbi stream of the segment
  bool tail = true;           // byte de-interleaver + RS + descrambler right behind (single period)
  long long avail = 0;        // samples in memory from iq[0] on (0: nsamples): a period of a longer segment is given a look-ahead window, the stream goes on behind it
  bool no_small = false;      // acq_only: not through acq_small_kernel (it has handed the period back: RxState.small_viol)
  bool tps_init = false;      // not a continuation, but the pilot engine's members in front of the first symbol are in the handle (TpsState: a piece of a stream whose
                              // counters are known): the segment-parallel bookkeeping starts from them instead of from blank ones
  bool cut_set = false;       // the cut of this period is given here instead of by the handle (walk_window: a cut belongs to the first period of its window) ...
  long long sym_off = 0; int delay = 0;   // ... dvbt_rx_cut.stream_symbol_offset, .start_delay_symbols
};

// samples at the OFDM elementary rate: the segment itself, or its resampled image (next row 2)
static int prepare_chain(dvbt_rx *h, const float2 *iq, size_t nsamples, hipStream_t s, const float2 **chain, size_t *chain_n)
{
  if (nsamples > h->max_samples) return fail(DVBT_ERR_CAPACITY, "segment longer than max_samples");
  *chain = iq; *chain_n = nsamples;
  if (h->rsd.ri) {   // the segment arrives at the file rate; resample + scale into the chain's input buffer first
    const long long cnt = (long long)(((unsigned long long)nsamples * h->rsd.ri + h->rsd.rd - 1) / h->rsd.rd);
    hipLaunchKernelGGL(resample_scale_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, iq, 0ll, (long long)nsamples, 0ll, cnt, h->rsd.ri, h->rsd.rd,
                       h->rsd.nt, (const float *)h->rsd.br, h->prm.front_scale == 0.f ? 1.0f : h->prm.front_scale, h->rs_iq);
    *chain = h->rs_iq; *chain_n = (size_t)cnt;
  }
  return DVBT_OK;
}

// The Viterbi stage's chunk size for a stream of at most max_vit decoded bytes.  Chosen per segment so that the wavefront count is a whole number of "rounds" of the resident
// wavefront slots (4 chunks per wavefront, V3_WAVES_PER_CU wavefronts per CU: 2 per SIMD): equal-length chunks then finish together instead of leaving a partial last round, and
// longer chunks amortise the warm-up + traceback overlap (V3_WARM + ntraceback - 1 windows per chunk).  Measured on 65 superframes: 3 rounds of ~2900-byte chunks beat 5 rounds
// of ~1700 (less overlap) and 1 round of ~8600 (the SIMD's arbiter favours the older of its two wavefronts, which then finishes long before the other:
// profiles/r02_viterbi_attribution.jsonl).  With the proof: a whole number of blocks of windows (a decoder and its predecessor then both stand at the top of a block at the
// chunk's first window).
static int vit_chunk_bytes(const dvbt_rx *h, long long max_vit)
{
  long long B = h->prm.viterbi_chunk_bytes;
  if (B <= 0) {
    const long long slots = (long long)h->ncu * V3_WAVES_PER_CU * 4;   // chunks resident at once  (ncu is read at create: nothing but stream operations inside enqueue, which may be under capture)
    long long rounds = (max_vit + slots * kVitMaxChunk - 1) / (slots * kVitMaxChunk);
    if (rounds < 1) rounds = 1;
    B = (max_vit + slots * rounds - 1) / (slots * rounds);
    if (B < 256) B = 256;
  }
  if (h->prm.viterbi_verify != VIT_PLAIN && !h->prm.soft_decision) B = std::min<long long>((B + V3_BLK - 1) / V3_BLK * V3_BLK, (1ll << 30) / V3_BLK * V3_BLK);
  return (int)B;
}

// A8 + A9 + energy_descramble over the segment's Viterbi stream.  words_fixed < 0: the word count is the device's (one period, no host round trip)
static int enqueue_tail(dvbt_rx *h, hipStream_t s, long long max_words, long long words_fixed)
{
  if (words_fixed >= 0) {   // several periods: the host has laid the stream out and knows its length; one small kernel patches the device-side plan
    const long long items = (words_fixed / 8) & ~1ll;
    hipLaunchKernelGGL(tail_patch_kernel, dim3(1), dim3(1), 0, s, h->st, items);
  }
  hipLaunchKernelGGL(deint_rs_kernel, dim3((unsigned)((max_words + 63) / 64)), dim3(64), 0, s, (const uint8_t *)h->vit, h->deint_tap, h->rs_out,
                     h->st, 0ll, 0, 0ll, h->T.rs_tables(), h->prm.rs_oracle_compat, &h->st->rs_fail, &h->st->rs_corr, h->rs_defer, &h->st->rs_list_n, h->rs_defer_cap,
                     h->rs_sync);
  // the few bad words of lightly loaded wavefronts, one wavefront each (the deint tap shows the words as received: patches go to the payload only)
  hipLaunchKernelGGL(rs_fix_kernel, dim3(512), dim3(64), 0, s, (const RsDefer *)h->rs_defer, (const int *)&h->st->rs_list_n, h->rs_defer_cap, h->rs_out,
                     h->T.rs_tables(), h->prm.rs_oracle_compat, &h->st->rs_fail, &h->st->rs_corr, h->rs_sync);
  if (h->prm.descramble) {
    hipLaunchKernelGGL(descramble_scan_kernel, dim3(1), dim3(1024), 0, s, (const uint8_t *)h->rs_out, h->st, h->descr_runs, h->descr_nruns,
                       (const unsigned long long *)h->rs_sync, h->cut.descr_call_phase - 1);
    hipLaunchKernelGGL(descramble_runs_kernel, dim3(1024), dim3(256), 0, s, (const uint8_t *)h->rs_out, (const uint8_t *)h->T.prbs,
                       (const RxState *)h->st, (const DescrRun *)h->descr_runs, (const int *)h->descr_nruns, h->ts_out);
  }
  return DVBT_OK;
}

// one lock period of the chain, iq at the OFDM elementary rate
static int enqueue(dvbt_rx *h, const float2 *iq, size_t nsamples, hipStream_t s, const EnqOpt &o = EnqOpt())
{
  const Dims &d = h->d;
  if (nsamples > h->chain_max) return fail(DVBT_ERR_CAPACITY, "segment longer than max_samples");
  if (nsamples < (size_t)(2 * d.N + d.cp + 16)) return fail(DVBT_ERR_INVALID, "segment shorter than one acquisition window");
  FrontParams fp = h->fp;
  const long long cut_sym_off = o.cut_set ? o.sym_off : (long long)h->cut.stream_symbol_offset;
  const int cut_delay = o.cut_set ? o.delay : h->cut.start_delay_symbols;
  fp.hunt_known = cut_delay > 0 ? 1 : 0;
  fp.si_start = cut_delay % 68; fp.fi_start = (d.fi_start + cut_delay / 68) % 4;     // dvbt_rx_cut.start_delay_symbols: the hunt fires that many symbols behind the true start
  fp.ncalls = (int)((nsamples - (2 * d.N + d.cp + 16)) / (d.N + d.cp) + 1);
  fp.hist = o.hist; fp.keep_last = o.keep_last ? 1 : 0; fp.avail = o.avail > 0 ? o.avail : (long long)nsamples;
  const int C = fp.ncalls, N = d.N;
  h->cur_stream = s;
  // dvbt_rx_params.front_priority: everything in front of the Viterbi decoder on the handle's high-priority stream (behind whatever the caller's stream holds), the
  // decoder and the tail back on the caller's stream
  hipStream_t const s_user = s;
  const bool split = h->front_stream && !o.acq_only && !o.skip_acq && !o.continuation && h->timing != 1 && C >= 512 && !h->prm.launch_graph;
  if (split) { HIPCHK(hipEventRecord(h->front_ev[0], s_user)); HIPCHK(hipStreamWaitEvent(h->front_stream, h->front_ev[0], 0)); s = h->front_stream; }
  auto join_front = [&]() -> int { if (split && s != s_user) { HIPCHK(hipEventRecord(h->front_ev[1], s)); s = s_user; HIPCHK(hipStreamWaitEvent(s, h->front_ev[1], 0)); } return DVBT_OK; };
  const bool tm = h->timing == 1 && !o.acq_only, tmv = h->timing != 0 && !o.acq_only;   // an event record costs ~6 us of an otherwise idle stream
  if (tm) HIPCHK(hipEventRecord(h->ev[ST_ACQ], s));
  if (o.skip_acq) {
    // what acq_init_fsm_kernel's reset would have done on top of the acq_only run's (tracker flags and the symbol ticket are still clear)
    if (!o.continuation && !o.tps_init) HIPCHK(hipMemsetAsync(h->tps_state, 0, sizeof(TpsState), s));
  } else {
  const AcqState *carry = nullptr;
  int tries = C < o.init_tries ? C : o.init_tries;
  // initial search: the metric of all `tries` windows in one launch (32 workgroups per window: four windows cost the latency of one), then ONE launch of the
  // state machine, which examines them in order and stops at the first peak -- window 0 on a stream that is there, so the later windows' metric is rarely
  // looked at; computing it unasked costs nothing on an otherwise idle device and saves two launches per lock period
  // (the FSM launch also clears the trackers' flag words, the symbol kernel's ticket and, for a period that starts the pilot engine afresh, its state)
  hipLaunchKernelGGL(acq_metric_kernel, dim3((N + 255) / 256, tries), dim3(256), 0, s, iq, fp, (const RxState *)h->st, 0, h->g_init, h->l_init, 0);
  const AcqReset rz = {h->trk_flags, h->sym_ticket, (o.acq_only || o.continuation || o.tps_init) ? nullptr : reinterpret_cast<int *>(h->tps_state), (int)(sizeof(TpsState) / 4),
                       o.carry_avg, o.use_carry ? 1 : 0};
  hipLaunchKernelGGL(acq_init_fsm_kernel, dim3(1), dim3(1024), (size_t)N * 5, s, fp, h->st, (const float2 *)h->g_init, (const float *)h->l_init, carry, 0, tries, rz);
  if (o.acq_only && !o.no_small && C <= ACQ_SMALL_MAX_CALLS) {
    // the lock-period walk's short look-ahead windows: everything behind the initial search in one launch, work in proportion to the symbols the lock holds
    const int cpc = acq_small_cpc(d.cp);
    h->n_small++;
    hipLaunchKernelGGL(acq_small_kernel, dim3(1), dim3(64 * (1 + cpc / 2)), (size_t)cpc * 2 * (d.cp + 2 * ACQ_R) * sizeof(float2), s, iq, fp, h->st, h->meta, cpc, h->drift.flags);
  } else {
  if (o.acq_only) h->n_general++;
  {   // where the tracking metric is computed: CP position predicted per call from coarse estimates every ACQ_ANCHOR calls (sample-clock drift)
    const int n_anchors = (C - 1) / ACQ_ANCHOR;
    if (n_anchors > 0) hipLaunchKernelGGL(acq_anchor_kernel, dim3(n_anchors), dim3(1024), acq_anchor_lds_bytes(N, d.cp), s, iq, fp, (const RxState *)h->st, h->anchor_pos);
    hipLaunchKernelGGL(acq_centre_kernel, dim3((C + 1023) / 1024), dim3(1024), 0, s, fp, (const RxState *)h->st, (const int *)h->anchor_pos, n_anchors, h->centre);
  }
  hipLaunchKernelGGL(acq_track_metric_kernel, dim3((C + ACQ_TM_CALLS - 1) / ACQ_TM_CALLS), dim3(256), 0, s, iq, fp, (const RxState *)h->st, (const int *)h->centre, h->g_trk, h->l_trk);
  constexpr int kIters = TRK_ROUNDS;            // Jacobi iterations of the window placement (one launch); flags[kIters] = need_seq
  hipLaunchKernelGGL(acq_track_fused_kernel, dim3((C + TRK_OWN - 1) / TRK_OWN), dim3(256), 0, s, fp, (const RxState *)h->st, (const float2 *)h->g_trk,
                     (const float *)h->l_trk, h->trk_cp_a, h->trk_eps, h->trk_flags, (const int *)h->centre);
  hipLaunchKernelGGL(acq_finalize_kernel, dim3((C + 1023) / 1024), dim3(1024), 0, s, fp, h->st, (const int *)h->trk_cp_a, (const float *)h->trk_eps,
                     (const int *)h->trk_flags, kIters - 1, h->meta, h->trk_flags + kIters, (const float *)h->l_trk, (const int *)h->centre);
  // not settled: the sequential walk, positions and epsilon only (flags[kIters] = need_seq); flags[kIters + 1] = need_heavy, which it raises for a period
  // outside the closed form of the phase -- that one goes through the float-faithful tracker of the block API
  hipLaunchKernelGGL(acq_track_light_kernel, dim3(1), dim3(64), 0, s, fp, h->st, (const float2 *)h->g_trk, (const float *)h->l_trk, h->meta,
                     (const int *)(h->trk_flags + kIters), h->trk_flags + kIters + 1, (const int *)h->centre, iq);
  hipLaunchKernelGGL(acq_track_kernel, dim3(1), dim3(64), 0, s, fp, h->st, (const float2 *)h->g_trk, (const float *)h->l_trk, h->meta,
                     (const int *)(h->trk_flags + kIters + 1), (AcqState *)nullptr, (const int *)h->centre, iq);
  }
  if (o.acq_only) {
    HIPCHK(hipMemcpyAsync(h->st_host, h->st, sizeof(RxState), hipMemcpyDeviceToHost, s));
    HIPCHK(hipGetLastError());
    return DVBT_OK;
  }
  }
  if (tm) HIPCHK(hipEventRecord(h->ev[ST_FFT], s));
  // A1 tail + A2 + A3 in one kernel: the FFT item of a symbol never leaves LDS (acq/fft taps are written only when enabled)
  // a period found by acq_small_kernel whose state block the host has just read back: the drift model's verdict is known (and flags[1] cleared on the device)
  const bool drift_off = o.skip_acq && h->st_host->drift_known_off;
  if (!drift_off) {
    // the wander of the reference's float phase accumulator (k_drift.hpp): tables per call, the fixed point (one workgroup), the deviations per 32-sample block;
    // every kernel returns at once when the lock period has no usable carrier offset (drift.flags, device side: no host round trip).  (Round 4 put the
    // launches behind drift_exact_kernel and the DRIFT instantiation on a second stream, beside the plain symbol kernel: its persistent workgroups hold every
    // VGPR of the machine, the empty launches waited for them to retire and the join cost what the fork had saved -- measured, taken out again.)
    const DriftBufs &D = h->drift;
    hipLaunchKernelGGL(drift_prep_kernel, dim3((C + 255) / 256), dim3(256), 0, s, fp, (const RxState *)h->st, (const SymMeta *)h->meta, D);
    hipLaunchKernelGGL(drift_exact_kernel, dim3(1), dim3(1024), 0, s, fp, (const RxState *)h->st, (const SymMeta *)h->meta, D);
    // three rounds of the fixed point (d ping-pongs between D.d and D.S), then the prefix sums S the table kernel reads
    const dim3 rg((C + 255) / 256);
    hipLaunchKernelGGL(drift_round_kernel<0>, rg, dim3(256), 0, s, fp, (const RxState *)h->st, (const SymMeta *)h->meta, D, (const double *)nullptr, D.d);
    hipLaunchKernelGGL(drift_round_kernel<1>, rg, dim3(256), 0, s, fp, (const RxState *)h->st, (const SymMeta *)h->meta, D, (const double *)D.d, D.S);
    hipLaunchKernelGGL(drift_round_kernel<1>, rg, dim3(256), 0, s, fp, (const RxState *)h->st, (const SymMeta *)h->meta, D, (const double *)D.S, D.d);
    hipLaunchKernelGGL(drift_round_kernel<2>, rg, dim3(256), 0, s, fp, (const RxState *)h->st, (const SymMeta *)h->meta, D, (const double *)D.d, D.S);
    hipLaunchKernelGGL(drift_table_kernel, dim3(C), dim3(N / 32 < 256 ? N / 32 : 256), 0, s, fp, (const RxState *)h->st, (const SymMeta *)h->meta, D);
  }
  const bool taps = h->acq_tap || h->fft_out || h->eq;
#define SYM_ARGS iq, fp, (const RxState *)h->st, (const SymMeta *)h->meta, (const float2 *)h->T.tw, h->acq_tap, h->fft_out, h->T.demod_tables(), h->eq, h->tpsval, h->info, \
                 h->T.inner_params(d.payload), (const float2 *)h->T.points, (const unsigned char *)h->T.label_tab, h->labels, h->sym_ticket, (const float *)h->drift.delta, (const int *)h->drift.flags, h->csi
  // 8k: persistent workgroups, two per CU (k_symbol8k.hpp); 2k: the same design, four symbols per workgroup (k_symbol2k.hpp).  The plain and the DRIFT
  // instantiation are both launched: the one that drift.flags[1] does not select returns before it takes a symbol
  // (the persistent workgroups take symbols off a ticket: a short lock period launches no more of them than it has symbols)
  const int g8 = std::min(h->sym_grid, C), g2 = std::min(h->sym_grid, (C + S2_Q - 1) / S2_Q);
  if (N == S8_N && !taps) {
    hipLaunchKernelGGL((symbol8k_kernel<false, false>), dim3(g8), dim3(S8_T), S8_LDS_BYTES, s, SYM_ARGS);
    if (!drift_off) hipLaunchKernelGGL((symbol8k_kernel<false, true>), dim3(g8), dim3(S8_T), S8_LDS_BYTES, s, SYM_ARGS);
  } else if (N == S8_N) {
    hipLaunchKernelGGL((symbol8k_kernel<true, false>), dim3(g8), dim3(S8_T), S8_LDS_BYTES, s, SYM_ARGS);
    if (!drift_off) hipLaunchKernelGGL((symbol8k_kernel<true, true>), dim3(g8), dim3(S8_T), S8_LDS_BYTES, s, SYM_ARGS);
  } else if (N == S2_N && !taps) {
    hipLaunchKernelGGL((symbol2k_kernel<false, false>), dim3(g2), dim3(S2_T * S2_Q), S2_LDS_BYTES, s, SYM_ARGS);
    if (!drift_off) hipLaunchKernelGGL((symbol2k_kernel<false, true>), dim3(g2), dim3(S2_T * S2_Q), S2_LDS_BYTES, s, SYM_ARGS);
  } else if (N == S2_N) {
    hipLaunchKernelGGL((symbol2k_kernel<true, false>), dim3(g2), dim3(S2_T * S2_Q), S2_LDS_BYTES, s, SYM_ARGS);
    if (!drift_off) hipLaunchKernelGGL((symbol2k_kernel<true, true>), dim3(g2), dim3(S2_T * S2_Q), S2_LDS_BYTES, s, SYM_ARGS);
  }
#undef SYM_ARGS
  if (tm) HIPCHK(hipEventRecord(h->ev[ST_DEMOD], s));
  if (!o.continuation) {
    hipLaunchKernelGGL(tps_vote_kernel, dim3((C + 63) / 64), dim3(256), 0, s, (const float2 *)h->tpsval, d.n_tps, (const RxState *)h->st, 0,
                       (const float2 *)nullptr, h->maj, fp.keep_last);
    // flags[8] = first superframe-start candidate (min), flags[9] = need_seq for the TPS bookkeeping (set by acq_init_fsm_kernel's reset)
    hipLaunchKernelGGL(tps_fsm_par_kernel, dim3((C + TPS_THREADS * TPS_SEG - 1) / (TPS_THREADS * TPS_SEG)), dim3(TPS_THREADS), 0, s, fp, (const RxState *)h->st, (const SymInfo *)h->info,
                       (const int *)h->maj, h->sym_index, h->tps_edges, h->trk_flags + 8, &h->st->tps_bits, (const unsigned short *)h->T.tps_bch,
                       o.tps_init ? (const TpsState *)h->tps_state : (const TpsState *)nullptr);
    hipLaunchKernelGGL(tps_tail_kernel, dim3(1), dim3(256), 0, s, fp, h->st, (const SymInfo *)h->info, (const int *)h->maj, h->tps_state, h->sym_index,
                       (const TpsEdge *)h->tps_edges, (const int *)(h->trk_flags + 8), h->trk_flags + 9, 0, h->vp, cut_sym_off);
  } else {
    // the pilot engine's members live on (FIFO, symbol and frame counters: reference_signals_impl.h); the sync_start tag on the period's first
    // item clears d_init (the superframe hunt starts over); DBPSK against the last symbol in front of the gap.  Sequential bookkeeping.
    hipLaunchKernelGGL(tps_vote_kernel, dim3((C + 63) / 64), dim3(256), 0, s, (const float2 *)h->tpsval, d.n_tps, (const RxState *)h->st, 0,
                       (const float2 *)h->tps_prev, h->maj, fp.keep_last);
    hipLaunchKernelGGL(tps_tail_kernel, dim3(1), dim3(256), 0, s, fp, h->st, (const SymInfo *)h->info, (const int *)h->maj, h->tps_state, h->sym_index,
                       (const TpsEdge *)nullptr, (const int *)nullptr, (int *)nullptr, 1, h->vp, cut_sym_off);
  }
  if (tm) HIPCHK(hipEventRecord(h->ev[ST_INNER], s));
  InnerParams ip = h->T.inner_params(d.payload);
  long long max_vit = (long long)C * d.payload * d.m * d.k / (8 * d.n) + 1;
  if (o.vit_off + (size_t)max_vit > h->vit_cap) return fail(DVBT_ERR_CAPACITY, "Viterbi stream buffer too small for the segment's lock periods");
  if (h->prm.soft_decision) {
    // soft decisions (k_soft.hpp): LLRs from the equalised carriers and their channel state, A5 + A6 as one gather on the soft values, soft-input decoder
    const float step = 2.0f * d.norm;
    hipLaunchKernelGGL(soft_demap_kernel, dim3(C), dim3(256), (size_t)d.payload * d.m, s, (const float2 *)h->eq, (const float *)h->csi, (const RxState *)h->st, ip, (const float2 *)h->T.points,
                       1.0f / (step * step), (const int *)h->sym_index, (const uint16_t *)h->soft_tab, h->soft_a);
    { int r = join_front(); if (r) return r; }
    if (tmv) HIPCHK(hipEventRecord(h->ev[ST_VIT], s));
    // the decoder: four chunks per wavefront, chunk size for whole rounds of the wavefront slots
    const S4Plan sp = s4_plan(max_vit, d.ntb);
    const unsigned grid = std::min(s4_grid(max_vit, d.ntb), h->soft_grid);   // never more slots than the scratch has (the kernel strides its tasks by gridDim)
    hipLaunchKernelGGL(viterbi_soft4_kernel, dim3(grid), dim3(64 * S4_WAVES), 0, s, (const int8_t *)h->soft_a, h->vit + o.vit_off, (const RxState *)h->st, h->vp,
                       h->soft_scratch, sp.B, sp.nsteps);
  } else {
  // A5 + A6 on the label bytes of the symbols from first_out on (A4 ran inside the symbol kernel)
  hipLaunchKernelGGL(inner_kernel<6>, dim3(C), dim3(INNER_THREADS), inner_lds_bytes((size_t)d.payload), s, (const float2 *)nullptr, (const uint8_t *)h->labels, ip,
                     (const RxState *)h->st, 0, (const int *)h->sym_index, (const float2 *)nullptr, (const unsigned char *)nullptr,
                     (const uint16_t *)h->T.H, (const uint16_t *)h->T.Hinv, (uint8_t *)nullptr, h->symdeint_tap, h->bitdeint, h->bitdeint_lp);
  { int r = join_front(); if (r) return r; }
  if (tmv) HIPCHK(hipEventRecord(h->ev[ST_VIT], s));
  VitParams vp = h->vp;
  vp.chunk_bytes = vit_chunk_bytes(h, max_vit);
  // (hierarchical modes: the decoder is handed bytes of which only two (HP) or m - 2 (LP) bits carry anything, unpacked as m coded bits each, viterbi_decoder_impl.cc:236-243 --
  // a degenerate input, two thirds constant zeros, whose survivors need not merge inside a chunk's warm-up: about one chunk start in 250 is not proven at 72 windows and is
  // decoded again by the repair pass.  Until round 6 these modes ran ONE decoder from the stream's start, on one wavefront.)
  const bool proof = h->vproof.snap && h->prm.viterbi_verify != VIT_PLAIN;
  V3Aux ax = h->vproof.aux();
  if (proof) {
    if (max_vit / vp.chunk_bytes + 2 > h->vproof.cap) return fail(DVBT_ERR_CAPACITY, "Viterbi proof: state buffer too small for this segment's chunks");
    h->vit_checked = true;
  }
  // hierarchical modes: the decoder reads the bit de-interleaver's output 0 (HP: what the flowgraphs connect), or its output 1 on request; it unpacks d_m
  // bits of every byte either way (viterbi_decoder_impl.cc:93,236-243: the reference's decoder knows no priority streams)
  launch_viterbi(s, (const uint8_t *)((h->prm.hier_stream && h->bitdeint_lp) ? h->bitdeint_lp : h->bitdeint), h->vit + o.vit_off, (const RxState *)h->st, 0ll, vp, 0ll, 0ll, max_vit,
                 proof ? &ax : nullptr, h->prm.viterbi_verify);
  }
  if (tmv) HIPCHK(hipEventRecord(h->ev[ST_RS], s));
  if (o.tail) { int r = enqueue_tail(h, s, max_vit / 204 + 1, -1); if (r) return r; }
  if (tm) HIPCHK(hipEventRecord(h->ev[ST_END], s));
  if (tmv) h->ev_recorded = true;
  HIPCHK(hipMemcpyAsync(h->st_host, h->st, sizeof(RxState), hipMemcpyDeviceToHost, s));
  HIPCHK(hipGetLastError());
  h->pending = true;
  return DVBT_OK;
}

extern "C" int dvbt_rx_segment_enqueue_device(dvbt_rx *h, const void *iq_device, size_t nsamples, void *stream)
{
  if (!h || !iq_device) return fail(DVBT_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->prm.device));
  hipStream_t s = stream ? (hipStream_t)stream : h->own_stream;
  const float2 *chain; size_t chain_n;
  int r = prepare_chain(h, (const float2 *)iq_device, nsamples, s, &chain, &chain_n); if (r) return r;
  h->n_periods = 1; h->seg_offset = 0;
  if (h->prm.launch_graph && !h->timing && !h->rsd.ri && !(h->acq_tap || h->fft_out || h->symdeint_tap || h->deint_tap)) {   // (not with the debug taps: a replay would write buffers that dvbt_rx_enable_taps(h, 0) frees)
    // the launch sequence as ONE graph launch: captured the first time this (segment, length, stream, cut) is seen
    for (auto &ge : h->graphs)
      if (ge.iq == chain && ge.n == chain_n && ge.s == s && ge.sym_off == h->cut.stream_symbol_offset && ge.delay == h->cut.start_delay_symbols && ge.phase == h->cut.descr_call_phase) {
        HIPCHK(hipGraphLaunch(ge.exec, s));
        h->cur_stream = s; h->pending = true;
        return DVBT_OK;
      }
    if (h->graphs.size() < 16) {
      hipGraph_t gr = nullptr; hipGraphExec_t ex = nullptr;
      HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      const int r = enqueue(h, chain, chain_n, s);
      const hipError_t e = hipStreamEndCapture(s, &gr);
      if (r) { if (gr) (void)hipGraphDestroy(gr); return r; }
      if (e != hipSuccess || !gr) return fail(DVBT_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
      const hipError_t e2 = hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
      (void)hipGraphDestroy(gr);
      if (e2 != hipSuccess) return fail(DVBT_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e2));
      h->graphs.push_back(dvbt_rx::GraphEntry{chain, chain_n, s, (long long)h->cut.stream_symbol_offset, h->cut.start_delay_symbols, h->cut.descr_call_phase, ex});
      HIPCHK(hipGraphLaunch(ex, s));
      h->cur_stream = s; h->pending = true;
      return DVBT_OK;
    }
  }
  return enqueue(h, chain, chain_n, s);
}

static void fill_report(dvbt_rx *h, const RxState &s, dvbt_rx_report &r)
{
  memset(&r, 0, sizeof r);
  r.status = s.status; r.n_symbols = s.n_symbols; r.first_out_symbol = s.first_out; r.n_out_symbols = s.n_out_symbols;
  r.cp_start0 = s.cp_start0; r.first_call = s.call0;
  // the call that loses the lock consumes half a window (to_consume / 2, ofdm_sym_acquisition_impl.cc:545-559): that half step is
  // what moves the search windows of the re-acquisition to a different phase of the symbol grid
  r.resume_sample = ((s.status & 2) && !(s.status & 1)) ? (int64_t)(s.call0 + s.n_symbols) * (h->d.N + h->d.cp) + (h->d.N + h->d.cp) / 2 : 0;
  r.n_viterbi_bytes = s.n_vit_bytes; r.n_rs_items = s.n_rs_items; r.n_rs_bytes = s.n_rs_words * 188;
  r.n_ts_bytes = s.n_ts_bytes; r.rs_fail_words = s.rs_fail; r.rs_corrected_symbols = s.rs_corr;
  r.stream_symbol_offset = s.sym_off; r.ts_first_packet = s.ts_first_packet; r.stream_rs_items = s.stream_rs_items;
  // TPS as received (reference_signals_impl.cc:883-916: fields are MSB first, s_i = bit i of the word)
  r.tps_valid = (int32_t)(s.tps_bits >> 63); r.tps_bits = s.tps_bits & ~(1ull << 63);
  if (r.tps_valid) {
    auto fld = [&](int first, int last) { int v = 0; for (int i = first; i <= last; i++) v = (v << 1) | (int)((s.tps_bits >> i) & 1ull); return v; };
    r.tps_length_indicator = fld(17, 22); r.tps_constellation = fld(25, 26); r.tps_hierarchy = fld(27, 29);
    r.tps_code_rate_hp = fld(30, 32); r.tps_code_rate_lp = fld(33, 35); r.tps_guard_interval = fld(36, 37);
    r.tps_transmission_mode = fld(38, 39); r.tps_cell_id = fld(40, 47);
    const dvbt_rx_params &q = h->prm;
    r.tps_mismatch = (r.tps_constellation != q.constellation ? 1 : 0) | (r.tps_hierarchy != q.hierarchy ? 2 : 0) | (r.tps_code_rate_hp != q.code_rate ? 4 : 0) |
                     (r.tps_guard_interval != q.guard_interval ? 8 : 0) | (r.tps_transmission_mode != q.transmission_mode ? 16 : 0);
    if (r.tps_mismatch) r.status |= 16;
  }
}

extern "C" int dvbt_rx_segment_finish(dvbt_rx *h, dvbt_rx_report *rep)
{
  if (!h) return fail(DVBT_ERR_INVALID, "null handle");
  if (!h->pending) return fail(DVBT_ERR_STATE, "no segment enqueued");
  HIPCHK(hipStreamSynchronize(h->cur_stream));
  h->pending = false;
  if (h->timing && h->ev_recorded) {
    for (int i = h->timing == 2 ? ST_VIT : 0; i < (h->timing == 2 ? ST_RS : ST_END); i++) { float ms = 0; if (hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]) == hipSuccess) h->acc_ms[i] += ms; }
    float ms = 0; if (h->timing == 1 && hipEventElapsedTime(&ms, h->ev[0], h->ev[ST_END]) == hipSuccess) h->acc_ms[ST_END] += ms;
    h->n_timed++; h->ev_recorded = false;
  }
  dvbt_rx_report r;
  fill_report(h, *h->st_host, r);
  r.n_lock_periods = r.first_out_symbol >= 0 ? 1 : 0;
  h->last = r; h->have_last = true;
  if (rep) *rep = r;
  return DVBT_OK;
}

// The synchronous entries follow the reference through every loss of the CP lock inside the segment (ofdm_sym_acquisition_impl.cc:545-559;
// oracle/o_chain.c restates the downstream consequences).  Phase A walks the segment with ofdm_sym_acquisition alone: where each lock period
// starts, how many symbols it holds, where the search resumes (half a window behind a lost lock; one window after each call that found
// nothing), d_avg carried along.  Phase B runs the chain over every period that acquired symbols, in order: TPS state carried, the period's last
// item kept when a later period follows, every period's Viterbi stream appended to the segment's at a multiple of two de-interleaver items.
// Then the byte de-interleaver, RS and the descrambler run once over the whole stream.
struct LockPeriod { size_t off; int n_symbols; float avg_in; bool carry; int call0, cp_start0; bool lost; };
static void set_ctx(dvbt_rx *h, int k)
{
  if (k != h->ctx && !h->graphs.empty()) drop_graphs(h);
  h->ctx = k; h->st = h->st_ctx[k]; h->meta = h->meta_ctx[k]; h->trk_flags = h->trk_flags_ctx[k]; h->sym_ticket = h->sym_ticket_ctx[k]; h->drift.flags = h->drift_flags_ctx[k];
  h->st_host = h->st_host_ctx[k];
}

static int segment_periods(dvbt_rx *h, const float2 *chain, size_t chain_n, hipStream_t s, dvbt_rx_report *rep)
{
  const Dims &d = h->d;
  const size_t L = (size_t)(d.N + d.cp), win = (size_t)(2 * d.N + d.cp + 16);
  std::vector<LockPeriod> per;
  h->plog.clear(); h->plog_bytes = 0;
  bool capped = false;                                            // the walk was cut short: the rest of the segment is not decoded (status bit 8)
  h->periods.clear();
  // what the decode of the periods accumulates
  struct Book {
    size_t acc = 0; int delivering = 0, processed = 0; bool any = false; dvbt_rx_report first_rep; RxState last_st; int total_symbols = 0; size_t last_off = 0;
  } bk, snap[2];
  memset(&bk.first_rep, 0, sizeof bk.first_rep); bk.first_rep.first_out_symbol = -1;
  memset(&bk.last_st, 0, sizeof bk.last_st); bk.last_st.first_out = -1; bk.last_st.status = 1;
  bool snap_assumed[2] = {false, false};                          // ... and were decoded on the assumption that a later period has items
  bool need_full = false;
  int ctx_last = -1;                                              // acquisition context of the last decoded period (-1: none yet): acquisitions go to the other one
  auto acq_ctx = [&]() -> int {                                   // switch to the context an acquisition may write (the other one may be in use by a decode in flight;
    const int k = ctx_last < 0 ? h->ctx : (ctx_last == 0 ? 1 : 0);   // what has to live through the periods of a segment -- the stream's TPS word -- is kept on the host)
                                                                  // (three contexts exist, the streaming entry's walk uses all of them: never ctx_last ^ 1, which is 3 for 2)
    if (k != h->ctx) set_ctx(h, k);
    return DVBT_OK;
  };
  // The decode of period p runs on the handle's second stream WHILE the caller's stream searches for period p + 1 (the search needs the samples and the average
  // the lost call left, nothing of the decode): a period costs the longer of the two instead of their sum.  The decode's outcome (its Viterbi byte count: where
  // the next period's bytes go; whether it delivered) is read before the NEXT decode is launched.
  hipStream_t s2 = h->aux_stream;
  HIPCHK(hipEventRecord(h->aux_ev, s)); HIPCHK(hipStreamWaitEvent(s2, h->aux_ev, 0));   // (whatever made the samples ready on the caller's stream)
  struct Pend { bool on = false; size_t p = 0, vit_off = 0; int ctx = 0; hipStream_t st = nullptr; } pend;
  unsigned long long tps_keep = 0;
  int snap_period[2] = {-1, -1}, sn = 0;                          // the last two decoded periods: snap[] and the device-side copies of the pilot engine's state were taken in front of
                                                                  // them (snap[sn]: the last one, snap[sn ^ 1]: the one before)
  // one period through the chain up to the Viterbi decoder.  later: a later period delivers items (the last item of this one leaves the demodulator too);
  // reuse: the acquisition results of the acq_only run just before are still in the handle (the period is decoded right behind its discovery)
  auto decode_finish = [&]() -> int {
    if (!pend.on) return DVBT_OK;
    HIPCHK(hipStreamSynchronize(pend.st));
    pend.on = false; h->pending = false;
    const size_t p = pend.p;
    const RxState &st = *h->st_host_ctx[pend.ctx];
    if (st.tps_bits) tps_keep = st.tps_bits;
    bk.processed++; bk.any = true; bk.last_st = st; bk.last_off = per[p].off;
    h->periods[p].first_out_symbol = 0;
    if (h->bd_log) {   // the period's decoder input, at a place that depends on the period alone (a period may be decoded twice: the entry is overwritten)
      size_t base = 0; for (size_t q = 0; q < p; q++) base += (size_t)std::max(per[q].n_symbols, 0);
      if (h->plog.size() <= p) h->plog.resize(p + 1, dvbt_period_tap{0, 0, 0, 0});
      dvbt_period_tap e = {(int64_t)(base * d.payload), 0, (int64_t)pend.vit_off, 0};
      if (st.first_out >= 0 && st.n_out_symbols > 0) {
        e.bitdeint_bytes = (int64_t)st.n_out_symbols * d.payload; e.viterbi_bytes = st.n_vit_bytes;
        HIPCHK(hipMemcpy(h->bd_log + e.bitdeint_offset, (h->prm.hier_stream && h->bitdeint_lp) ? h->bitdeint_lp : h->bitdeint, (size_t)e.bitdeint_bytes, hipMemcpyDeviceToDevice));
        h->plog_bytes = std::max(h->plog_bytes, (size_t)(e.bitdeint_offset + e.bitdeint_bytes));
      }
      h->plog[p] = e;
    }
    if (st.first_out >= 0) {
      h->periods[p].first_out_symbol = st.first_out + 1;
      if (bk.delivering == 0) { fill_report(h, st, bk.first_rep); bk.first_rep.segment_offset = (int64_t)per[p].off; }
      bk.acc = pend.vit_off + (size_t)st.n_vit_bytes; bk.delivering++;
    }
    return DVBT_OK;
  };
  auto decode_launch = [&](size_t p, bool later, bool reuse, hipStream_t ds) -> int {
    const int usable = per[p].n_symbols - (later ? 0 : 1);       // items that leave the demodulator
    bk.total_symbols += per[p].n_symbols;
    if (usable < 1) return DVBT_OK;
    if (!reuse) { int r = acq_ctx(); if (r) return r; }
    EnqOpt o; o.use_carry = per[p].carry; o.carry_avg = per[p].avg_in; o.hist = (long long)per[p].off; o.avail = (long long)(chain_n - per[p].off); o.continuation = bk.processed > 0; o.keep_last = later; o.tail = false; o.skip_acq = reuse;
    o.vit_off = bk.delivering > 0 ? (bk.acc / 3264) * 3264 : 0;   // convolutional_deinterleaver_impl.cc:109-120: the tag realigns the input
    o.init_tries = std::min(ACQ_INIT_TRIES_MAX, std::max(ACQ_INIT_TRIES, per[p].call0 + 1));   // (a second acquisition must reach the window the wide search found the peak in)
    // a period that ends in a lost lock is decoded over its own calls and the one that lost the lock, not over the whole rest of the segment
    size_t span = chain_n - per[p].off;
    if (per[p].lost) span = std::min(span, win + (size_t)(per[p].call0 + per[p].n_symbols) * L);
    int r = enqueue(h, chain + per[p].off, span, ds, o); if (r) return r;
    // the TPS carriers of the last demodulated symbol are the DBPSK reference of the next period's first one
    HIPCHK(hipMemcpyAsync(h->tps_prev, h->tpsval + (size_t)(usable - 1) * d.n_tps, sizeof(float2) * d.n_tps, hipMemcpyDeviceToDevice, ds));
    pend.on = true; pend.p = p; pend.vit_off = o.vit_off; pend.ctx = h->ctx; pend.st = ds;
    ctx_last = h->ctx;
    return DVBT_OK;
  };
  auto decode = [&](size_t p, bool later, bool reuse) -> int {     // the synchronous form (periods decoded again behind the walk)
    int r = decode_finish(); if (r) return r;
    r = decode_launch(p, later, reuse, s); if (r) return r;
    return decode_finish();
  };
  {   // ---- the walk: find a period (acquisition alone over a growing window), decode it, go on behind the call that lost the lock
    size_t off = 0; bool carry = false; float avg = 0.f; int fails = 0, guard = 0;
    EnqOpt so; size_t slook = 0;                                   // the search in flight on the caller's stream
    // the search for the next period is ENQUEUED before the host turns to the launches of the last period's decode: the device looks for period p + 1 while
    // the host is busy with the ~15 launches of period p's decode and while that decode runs
    auto launch_search = [&]() -> int {
      { int r = acq_ctx(); if (r) return r; }
      so = EnqOpt(); so.acq_only = true; so.use_carry = carry; so.carry_avg = avg; so.hist = (long long)off; so.avail = (long long)(chain_n - off);
      // searches that found nothing are followed by wider ones (4, 8, ... 64 windows per launch): dead air costs a launch sequence per 64 windows, not per 4
      so.init_tries = std::min(ACQ_INIT_TRIES_MAX, ACQ_INIT_TRIES << std::min(fails, 4));
      // the search and the tracker look at a window of the rest of the segment that grows while the lock holds to its end: a segment with many lock
      // periods costs its length a few times over, not its length times the number of periods (a call's outcome depends on the samples before it only)
      // (first window: 768 calls, or four times the previous period's length when the lock is being lost every few dozen symbols -- the metric, anchor and
      // tracker launches cost in proportion to the window)
      size_t look_calls = 767;
      if (!per.empty()) look_calls = std::min<size_t>(767, std::max<size_t>(47, 4 * (size_t)std::max(per.back().n_symbols, 0)));
      slook = std::min(chain_n - off, win + look_calls * L);
      // the first attempts take the whole rest of the segment: a lock that holds to its end (the usual case, possibly behind a start-up transient of a
      // few symbols) is one pass; only a stream that keeps losing the lock goes over to the short windows
      if (per.size() < 3 && guard < 8) slook = chain_n - off;
      return enqueue(h, chain + off, slook, s, so);
    };
    auto complete_search = [&]() -> int {
      for (;;) {
        HIPCHK(hipStreamSynchronize(s));
        if (!(h->st_host->status & 1) && h->st_host->small_viol && !so.no_small) { so.no_small = true; }   // outside acq_small_kernel's closed form: the general kernels
        else if ((h->st_host->status & 3) || slook >= chain_n - off) return DVBT_OK;
        else slook = std::min(chain_n - off, win + 4 * (slook - win) + 3 * L);
        int r = enqueue(h, chain + off, slook, s, so); if (r) return r;
      }
    };
    bool searching = off + win <= chain_n;
    if (searching) { int r = launch_search(); if (r) return r; }
    while (searching) {
      if (guard++ >= 4096) { capped = true; HIPCHK(hipStreamSynchronize(s)); break; }
      { int r = complete_search(); if (r) return r; }
      const RxState st = *h->st_host;
      const int tries = (int)std::min<size_t>((size_t)so.init_tries, (slook - win) / L + 1);      // what enqueue() examined
      if (st.status & 1) {                                       // no peak in these windows: the reference consumes them one by one and searches on
        off += (size_t)tries * L; avg = st.avg; carry = true; fails++;
        searching = off + win <= chain_n;
        if (searching) { int r = launch_search(); if (r) return r; }   // (the same context again: ctx_last has not moved)
        continue;
      }
      fails = 0;
      const int ctx_found = h->ctx;                               // the period's acquisition results live here until its decode has run
      { int r = decode_finish(); if (r) return r; }               // the decode of the period before has had this search's time
      // a period with items behind one that was decoded as the last one (the guess near the segment's end, below): everything is decoded again in order
      if (st.n_symbols >= 1 && snap_period[sn] >= 0 && !snap_assumed[sn]) need_full = true;
      const bool lost = (st.status & 2) != 0;
      per.push_back(LockPeriod{off, st.n_symbols, avg, carry, st.call0, st.cp_start0, lost});
      h->periods.push_back(dvbt_lock_period{(int64_t)off, st.call0, st.cp_start0, st.n_symbols, 0});
      // Decoded at once, on the acquisition results that are still in the handle.  Whether the period's last item leaves the demodulator depends on a
      // LATER period having items: a period that ends in a lost lock is decoded as if one did (corrected behind the walk if none does); a lock that
      // holds to the segment's end -- or is lost only where the samples run out -- makes the last period.
      // a lock that is lost where the samples run out (no room for another window behind it) ends the walk like one that holds to the end
      const size_t behind = off + (size_t)(st.call0 + st.n_symbols) * L + L / 2;
      const bool final_period = !lost || behind + win > chain_n;
      // ... and with less than 16 symbols left behind the first or second loss of a segment a later period with items is the less likely outcome (a stream
      // that simply ends; where the lock is being lost all the time, another short period is the likely one)
      const bool later_guess = !final_period && !(per.size() <= 2 && behind + win + 16 * L > chain_n);
      const size_t pidx = per.size() - 1;
      // the next search starts now, in the other context (free: the decode that used it has just been read back)
      searching = !final_period && per.size() < 1024;
      if (!final_period && per.size() >= 1024) capped = true;
      const bool will_decode = st.n_symbols - (later_guess ? 0 : 1) >= 1;
      if (searching) {
        off = behind; avg = st.avg_lost; carry = true;
        // a period that is going to be decoded keeps its context: the search takes the other one (that of the period decoded before: its decode has been read
        // back, the new period's takes its place as the last one).  A period that launches no decode leaves everything as it is
        const int keep = ctx_last; if (will_decode) ctx_last = ctx_found;
        int r = launch_search(); if (r) return r;
        ctx_last = keep;
      }
      const int ctx_search = h->ctx;
      set_ctx(h, ctx_found);
      if (will_decode) {                                         // keep the state in front of it
        sn ^= 1; snap[sn] = bk; snap_period[sn] = (int)pidx; snap_assumed[sn] = later_guess;
        HIPCHK(hipMemcpyAsync(h->tps_snap[sn], h->tps_state, sizeof(TpsState), hipMemcpyDeviceToDevice, s2));
        HIPCHK(hipMemcpyAsync(h->tps_prev_snap[sn], h->tps_prev, sizeof(float2) * d.n_tps, hipMemcpyDeviceToDevice, s2));
      }
      { int r = decode_launch(pidx, later_guess, true, s2); if (r) return r; }   // ... and runs while that search does
      set_ctx(h, ctx_search);
    }
    { int r = decode_finish(); if (r) return r; }
    int last_items = -1;                                          // the last period that has items at all
    for (size_t q = 0; q < per.size(); q++) if (per[q].n_symbols >= 1) last_items = (int)q;
    // The byte de-interleaver, the RS decoder and the report read the device-side state of the LAST decoded period, and that period's last item depends on
    // whether a later one has items.  Both are in order when the last period's lock held to the segment's end (the common case: its decode was the last
    // launch, or the lock was lost where the samples run out).  Behind a last decoded period that ended in a lost lock, searches have followed: they ran in
    // the other acquisition context (acq_ctx), so its state block and tracker results are intact and the handle just switches back to them.  Only when
    // the guess about a later period with items was wrong is the period decoded again from the state kept in front of it, acquisition included.  Should
    // that leave nothing to decode (a single item, not delivered after all), the period decoded before it is decoded again instead (the state in front of
    // the last TWO is kept; its context has been overwritten by the searches behind the last one).
    auto again = [&](int k) -> int {                              // period snap_period[k] once more, from the state in front of it, acquisition included
      const size_t z = (size_t)snap_period[k];
      bk = snap[k];
      HIPCHK(hipMemcpyAsync(h->tps_state, h->tps_snap[k], sizeof(TpsState), hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemcpyAsync(h->tps_prev, h->tps_prev_snap[k], sizeof(float2) * d.n_tps, hipMemcpyDeviceToDevice, s));
      h->periods[z].first_out_symbol = 0;
      int r = decode(z, last_items > (int)z, false); if (r) return r;
      for (size_t q = z + 1; q < per.size(); q++) bk.total_symbols += per[q].n_symbols;
      return DVBT_OK;
    };
    auto full = [&]() -> int {                                    // every period again in order, each with its own acquisition and the true `later`
      bk = Book(); memset(&bk.first_rep, 0, sizeof bk.first_rep); bk.first_rep.first_out_symbol = -1;
      memset(&bk.last_st, 0, sizeof bk.last_st); bk.last_st.first_out = -1; bk.last_st.status = 1;
      for (size_t q = 0; q < per.size(); q++) { h->periods[q].first_out_symbol = 0; int rr = decode(q, last_items > (int)q, false); if (rr) return rr; }
      return DVBT_OK;
    };
    const bool guess_ok = snap_period[sn] < 0 || snap_assumed[sn] == (last_items > snap_period[sn]);
    if (need_full) { ctx_last = -1; int r = full(); if (r) return r; }
    else if (snap_period[sn] >= 0 && !guess_ok) {
      ctx_last = -1;                                               // nothing to protect: the period is decoded again
      const int before = snap[sn].processed;
      { int r = again(sn); if (r) return r; }
      if (bk.processed == before && before > 0) {
        // the last period's single item is not delivered after all: the period decoded before it is the last one, and its device-side state is stale too
        if (snap_period[sn ^ 1] >= 0) { int r = again(sn ^ 1); if (r) return r; }
        else { int r = full(); if (r) return r; }                // (cannot happen: `before` > 0 means an earlier period was decoded)
      }
    }
  }
  if (ctx_last >= 0 && ctx_last != h->ctx) set_ctx(h, ctx_last);   // the searches behind the last decoded period ran in the other context
  size_t acc = bk.acc; const int delivering = bk.delivering, processed = bk.processed; const bool any = bk.any;
  const dvbt_rx_report first_rep = bk.first_rep; const RxState last_st = bk.last_st; const int total_symbols = bk.total_symbols; const size_t last_off = bk.last_off;
  dvbt_rx_report r;
  if (!any) {   // nothing was acquired (or single symbols only): report the last acquisition attempt
    RxState st = *h->st_host; st.first_out = -1; st.n_out_symbols = 0; st.n_vit_bytes = 0; st.n_rs_items = 0; st.n_rs_words = 0; st.n_ts_bytes = 0;
    st.rs_fail = st.rs_corr = 0; st.tps_bits = 0; st.ts_first_packet = 0; st.stream_rs_items = 0; st.status |= 4;
    fill_report(h, st, r);
    if (capped) r.status |= 256;
    h->n_periods = 0; h->seg_offset = 0;
    h->last = r; h->have_last = true; if (rep) *rep = r;
    return DVBT_OK;
  }
  if (delivering <= 1 && processed == 1 && per.size() >= 1 && acc == (size_t)last_st.n_vit_bytes) {
    // one period: the device-side plan of that period is already the segment's (cut-stream roundings included)
    int rr = enqueue_tail(h, s, (long long)(acc / 204 + 1), -1); if (rr) return rr;
  } else {
    int rr = enqueue_tail(h, s, (long long)(acc / 204 + 1), (long long)(acc / 204)); if (rr) return rr;
  }
  HIPCHK(hipMemcpyAsync(h->st_host, h->st, sizeof(RxState), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  RxState fin = *h->st_host;
  if (!fin.tps_bits) fin.tps_bits = tps_keep;                     // (the stream's TPS word: a period that saw no whole frame does not take it away)
  if (delivering > 1 || processed > 1) fin.n_vit_bytes = (long long)acc;
  fill_report(h, fin, r);
  // the front-end fields describe the LAST processed period (the debug taps hold it); the stream fields the whole segment
  r.segment_offset = (int64_t)last_off;
  if (r.resume_sample) r.resume_sample += (int64_t)last_off;
  r.n_lock_periods = delivering; r.total_symbols = total_symbols;
  if (capped) r.status |= 256;
  if (h->cut.stream_symbol_offset != 0 && (delivering > 1 || processed > 1)) r.status |= 128;   // a cut piece that lost the lock: laid out from its own start, not in the stream's coordinates
  if (delivering == 0) { r.first_out_symbol = -1; r.status |= 4; }
  h->n_periods = delivering; h->seg_offset = last_off;
  h->last = r; h->have_last = true;
  if (rep) *rep = r;
  return DVBT_OK;
}

// ---- one WINDOW of a walk (the streaming entry, dvbt_stream.inc): the lock-period walk of segment_periods over a stretch of a stream that goes on behind it,
// with what the reference's blocks hold between two work() calls carried in and out.  Two phases: A finds the window's lock periods with the acquisition alone
// (as segment_periods does), B decodes them in order up to the Viterbi decoder, every one with its own acquisition and with the TRUE answer to "does a later
// period deliver an item" (the period's last item leaves the demodulator only then, demod_reference_signals_impl.cc:88-94) -- no guess, nothing decoded twice.
// partial: the stream goes on.  The last period that has items is not final then (its end, and whether its last item is delivered, lie in samples to come): it is
// left to the next window, which starts its search where that period's search started, with that search's carried average.  The exception is an OPEN period (the
// lock holds to the window's end) whose superframe start lies at least `establish_calls` calls before the window's last call: it is decoded and kept -- the stream
// goes over to pieces from it (the head of an epoch).
// A8, A9 and the descrambler are the caller's (walk flush): the window only appends to the handle's Viterbi stream.
struct WalkPeriod {
  size_t off = 0; int n_symbols = 0, call0 = 0, cp_start0 = 0; float avg_in = 0.f; bool carry = false, lost = false;
  size_t behind = 0; float avg_lost = 0.f;                           // where the reference searches on after this period, and with which average
  bool decoded = false; int first_out = -1; int n_out_symbols = 0; size_t vit_off = 0; long long n_vit_bytes = 0;
};
struct WalkIO {
  // in
  bool partial = false;
  bool carry = false; float avg = 0.f;                               // ofdm_sym_acquisition: d_avg of the call that lost the lock before this window
  long long hist = 0;                                                // samples of the stream in memory in front of chain[0]
  bool continuation = false;                                         // the pilot engine has processed items before this window (TpsState / tps_prev are in the handle)
  bool tps_preset = false;                                           // it has not, but the members a chain in stable lock holds in front of the window are in the handle (TpsState)
  size_t acc = 0; int delivering = 0;                                // the walk's Viterbi stream so far (bytes in h->vit), periods that delivered so far
  bool cut = false; long long cut_sym_off = 0; int cut_delay = 0; size_t vit_pad = 0;   // the first decoded period continues a cut stream; its bytes go to h->vit + vit_pad
  long long establish_calls = 0;
  // out
  std::vector<WalkPeriod> per;
  size_t next_off = 0; bool next_carry = false; float next_avg = 0.f;   // partial, not established: the next window's search starts here (offset in this window)
  bool established = false; int head = -1; RxState head_st; long long head_ncalls = 0;   // calls of the head's grid that the window holds
  bool any_decoded = false; int status_or = 0;
  int total_symbols = 0;
};

static int walk_window(dvbt_rx *h, const float2 *chain, size_t chain_n, hipStream_t s, WalkIO &io)
{
  const Dims &d = h->d;
  const size_t L = (size_t)(d.N + d.cp), win = (size_t)(2 * d.N + d.cp + 16);
  const size_t maxn = h->chain_max;                                  // what one launch sequence of the handle can take: a longer window is looked at through this much
  std::vector<WalkPeriod> &per = io.per;
  per.clear(); h->periods.clear();
  io.established = false; io.head = -1; io.any_decoded = false; io.status_or = 0; io.total_symbols = 0;
  // One pass, three acquisition contexts: the period found last that has items is KEPT undecoded (whether its last item leaves the demodulator depends on a later
  // period having items, demod_reference_signals_impl.cc:88-94) -- when the next one with items is found, the kept one is decoded (later = true, on the acquisition
  // results that are still in its context, on the handle's second stream) while the caller's stream already searches for the one after: a period costs the longer
  // of search and decode, every acquisition runs once, nothing is guessed.
  hipStream_t s2 = h->aux_stream;
  HIPCHK(hipEventRecord(h->aux_ev, s)); HIPCHK(hipStreamWaitEvent(s2, h->aux_ev, 0));
  size_t acc = io.acc; int delivering = io.delivering; bool processed = io.continuation; bool cut_pending = io.cut;
  struct Pend { bool on = false; size_t p = 0, vit_off = 0; int ctx = 0; hipStream_t st = nullptr; } pend;
  std::vector<int> pctx;                                             // context of every period's acquisition
  int kept = -1;                                                     // the period that waits for its successor
  auto decode_finish = [&]() -> int {
    if (!pend.on) return DVBT_OK;
    HIPCHK(hipStreamSynchronize(pend.st));
    pend.on = false; h->pending = false;
    WalkPeriod &w = per[pend.p];
    const RxState &st = *h->st_host_ctx[pend.ctx];
    if (st.n_symbols != w.n_symbols || st.call0 != w.call0) io.status_or |= 512;
    processed = true; cut_pending = false; io.any_decoded = true;
    w.decoded = true; w.first_out = st.first_out; w.n_out_symbols = st.n_out_symbols; w.vit_off = pend.vit_off; w.n_vit_bytes = st.first_out >= 0 ? st.n_vit_bytes : 0;
    io.status_or |= st.status & ~(1 | 2 | 4);
    h->periods[pend.p].first_out_symbol = st.first_out >= 0 ? st.first_out + 1 : 0;
    if (st.first_out >= 0) { acc = pend.vit_off + (size_t)st.n_vit_bytes; delivering++; }
    io.head_st = st;
    return DVBT_OK;
  };
  // period p from the acquisition results in its context (the handle points at another one meanwhile: switched for the launches, switched back)
  auto decode_launch = [&](size_t p, bool later, hipStream_t ds) -> int {
    WalkPeriod &w = per[p];
    if (p != 0) cut_pending = false;                                 // a cut belongs to the lock period that reaches into the window from the stream before it
    const int usable = w.n_symbols - (later ? 0 : 1);                // items that leave the demodulator
    if (usable < 1) return DVBT_OK;
    const int back = h->ctx;
    set_ctx(h, pctx[p]);
    EnqOpt o; o.use_carry = w.carry; o.carry_avg = w.avg_in; o.hist = io.hist + (long long)w.off; o.avail = (long long)(chain_n - w.off);
    o.continuation = processed; o.keep_last = later; o.tail = false; o.skip_acq = true;
    o.tps_init = !processed && io.tps_preset;
    o.cut_set = true; o.sym_off = cut_pending ? io.cut_sym_off : 0; o.delay = cut_pending ? io.cut_delay : 0;
    // convolutional_deinterleaver_impl.cc:109-120: the tag realigns the block's input to a pair of items (3264 bytes of the walk's stream: h->vit[0] is a multiple of
    // 3264 bytes into it, the caller sees to that with vit_pad and by compacting in such multiples)
    o.vit_off = delivering > 0 ? (acc / 3264) * 3264 : (cut_pending ? io.vit_pad : acc);
    size_t span = std::min(chain_n - w.off, maxn);
    if (w.lost) span = std::min(span, win + (size_t)(w.call0 + w.n_symbols) * L);
    int r = enqueue(h, chain + w.off, span, ds, o);
    if (!r) { hipError_t e = hipMemcpyAsync(h->tps_prev, h->tpsval + (size_t)(usable - 1) * d.n_tps, sizeof(float2) * d.n_tps, hipMemcpyDeviceToDevice, ds); if (e != hipSuccess) r = fail(DVBT_ERR_HIP, "tps_prev copy"); }
    set_ctx(h, back);
    if (r) return r;
    pend.on = true; pend.p = p; pend.vit_off = o.vit_off; pend.ctx = pctx[p]; pend.st = ds;
    return DVBT_OK;
  };
  auto free_ctx = [&]() -> int {                                      // a context that neither the kept period nor the decode in flight lives in
    for (int k = 0; k < dvbt_rx::NCTX; k++) if (!(kept >= 0 && pctx[(size_t)kept] == k) && !(pend.on && pend.ctx == k)) return k;
    return 0;
  };
  // ---- the walk
  size_t off = 0; bool carry = io.carry; float avg = io.avg; int fails = 0, guard = 0; bool open = false;
  EnqOpt so; size_t slook = 0, srest = 0;
  auto launch_search = [&]() -> int {
    set_ctx(h, free_ctx());
    so = EnqOpt(); so.acq_only = true; so.use_carry = carry; so.carry_avg = avg; so.hist = io.hist + (long long)off; so.avail = (long long)(chain_n - off);
    so.init_tries = std::min(ACQ_INIT_TRIES_MAX, ACQ_INIT_TRIES << std::min(fails, 4));
    size_t look_calls = 767;
    if (!per.empty()) look_calls = std::min<size_t>(767, std::max<size_t>(47, 4 * (size_t)std::max(per.back().n_symbols, 0)));
    srest = std::min(chain_n - off, maxn);
    slook = std::min(srest, win + look_calls * L);
    if (per.size() < 3 && guard < 8) slook = srest;
    return enqueue(h, chain + off, slook, s, so);
  };
  auto complete_search = [&]() -> int {
    for (;;) {
      HIPCHK(hipStreamSynchronize(s));
      if (!(h->st_host->status & 1) && h->st_host->small_viol && !so.no_small) so.no_small = true;
      else if ((h->st_host->status & 3) || slook >= srest) return DVBT_OK;
      else slook = std::min(srest, win + 4 * (slook - win) + 3 * L);
      int r = enqueue(h, chain + off, slook, s, so); if (r) return r;
    }
  };
  bool searching = off + win <= chain_n;
  if (searching) { int r = launch_search(); if (r) return r; }
  while (searching) {
    if (guard++ >= 8192 || per.size() >= 4096) { io.status_or |= 256; HIPCHK(hipStreamSynchronize(s)); break; }
    { int r = complete_search(); if (r) return r; }
    const RxState st = *h->st_host;
    const int tries = (int)std::min<size_t>((size_t)so.init_tries, (slook - win) / L + 1);
    if (st.status & 1) {
      off += (size_t)tries * L; avg = st.avg; carry = true; fails++;
      searching = off + win <= chain_n;
      if (searching) { int r = launch_search(); if (r) return r; }
      continue;
    }
    fails = 0;
    WalkPeriod p; p.off = off; p.n_symbols = st.n_symbols; p.call0 = st.call0; p.cp_start0 = st.cp_start0; p.avg_in = avg; p.carry = carry; p.lost = (st.status & 2) != 0;
    p.behind = off + (size_t)(st.call0 + st.n_symbols) * L + L / 2; p.avg_lost = st.avg_lost;
    per.push_back(p); pctx.push_back(h->ctx);
    h->periods.push_back(dvbt_lock_period{(int64_t)off, st.call0, st.cp_start0, st.n_symbols, 0});
    io.total_symbols += st.n_symbols;
    if (!p.lost && srest < chain_n - off && !io.partial) io.status_or |= 256;   // (a final window longer than the handle's capacity: cannot happen with the stream's sizing)
    const bool last_period = !p.lost || p.behind + win > chain_n;      // the lock holds to the window's end, or is lost where its samples run out
    const size_t pidx = per.size() - 1;
    int to_decode = -1;
    if (p.n_symbols >= 1) {
      { int r = decode_finish(); if (r) return r; }                    // (its context becomes free, its byte count is where the next decode's bytes go)
      to_decode = kept; kept = (int)pidx;                              // the period kept so far has a successor with items: it is decoded now
    }
    if (last_period) { open = true; searching = false; }
    else {
      off = p.behind; avg = p.avg_lost; carry = true;
      searching = off + win <= chain_n;
    }
    // the kept period's decode is LAUNCHED behind the next search (the host's ~15 launches ride on the search's device time), in its own context
    if (to_decode >= 0 && searching) {
      pend.on = true; pend.ctx = pctx[(size_t)to_decode];              // (reserve its context for free_ctx)
      int r = launch_search(); pend.on = false; if (r) return r;
    } else if (searching) { int r = launch_search(); if (r) return r; }
    if (to_decode >= 0) { int r = decode_launch((size_t)to_decode, true, s2); if (r) return r; }
  }
  { int r = decode_finish(); if (r) return r; }
  // ---- the period that is still kept: the last one with items
  const int zl = kept;
  if (!io.partial) {
    if (zl >= 0) { int r = decode_launch((size_t)zl, false, s); if (r) return r; r = decode_finish(); if (r) return r; }
  } else {
    // (a period that continues a cut stream is never a head: its roundings are the old epoch's)
    const bool head_cand = zl >= 0 && zl == (int)per.size() - 1 && open && !per[zl].lost && !(cut_pending && zl == 0);
    if (head_cand) {
      const long long ncalls = (long long)((std::min(chain_n - per[zl].off, maxn) - win) / L) + 1;
      io.head_ncalls = ncalls;
      if (ncalls - per[zl].call0 >= io.establish_calls) {
        // keep what the pilot engine holds in front of the period: it is handed to the next window when the superframe start comes too late for the stream to go over to pieces
        HIPCHK(hipMemcpyAsync(h->tps_snap[0], h->tps_state, sizeof(TpsState), hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(h->tps_prev_snap[0], h->tps_prev, sizeof(float2) * d.n_tps, hipMemcpyDeviceToDevice, s));
        const size_t acc0 = acc; const int del0 = delivering; const bool proc0 = processed, cut0 = cut_pending; const bool any0 = io.any_decoded;
        { int r = decode_launch((size_t)zl, false, s); if (r) return r; r = decode_finish(); if (r) return r; }
        const WalkPeriod &w = per[zl];
        if (w.decoded && w.first_out >= 0 && ncalls - (w.call0 + w.first_out) >= io.establish_calls) { io.established = true; io.head = zl; set_ctx(h, pctx[(size_t)zl]); }
        else {
          acc = acc0; delivering = del0; processed = proc0; cut_pending = cut0; io.any_decoded = any0;
          per[zl].decoded = false;
          HIPCHK(hipMemcpyAsync(h->tps_state, h->tps_snap[0], sizeof(TpsState), hipMemcpyDeviceToDevice, s));
          HIPCHK(hipMemcpyAsync(h->tps_prev, h->tps_prev_snap[0], sizeof(float2) * d.n_tps, hipMemcpyDeviceToDevice, s));
          HIPCHK(hipStreamSynchronize(s));
        }
      }
    }
    if (!io.established) {
      if (zl >= 0) { io.next_off = per[zl].off; io.next_carry = per[zl].carry; io.next_avg = per[zl].avg_in; }
      else if (open && !per.empty()) { io.next_off = per.back().behind; io.next_carry = true; io.next_avg = per.back().avg_lost; }   // (a lock lost at once where the samples run out)
      else { io.next_off = off; io.next_carry = carry; io.next_avg = avg; }
      if (io.next_off > chain_n) io.next_off = chain_n;
      // the periods from zl on belong to the next window
      if (zl >= 0) { for (size_t q = (size_t)zl; q < per.size(); q++) io.total_symbols -= per[q].n_symbols; per.resize((size_t)zl); h->periods.resize((size_t)zl); }
    }
  }
  io.acc = acc; io.delivering = delivering; io.continuation = processed; io.cut = cut_pending;
  if (processed) io.tps_preset = false;
  h->n_periods = delivering; h->seg_offset = 0;
  return DVBT_OK;
}

extern "C" int dvbt_rx_segment_run(dvbt_rx *h, const void *iq_host, size_t nsamples, dvbt_rx_report *rep)
{
  if (!h || !iq_host) return fail(DVBT_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->prm.device));
  if (nsamples > h->max_samples) return fail(DVBT_ERR_CAPACITY, "segment longer than max_samples");
  if (!h->d_iq) HIPCHK(hipMalloc((void **)&h->d_iq, sizeof(float2) * h->max_samples));
  HIPCHK(hipMemcpyAsync(h->d_iq, iq_host, sizeof(float2) * nsamples, hipMemcpyHostToDevice, h->own_stream));
  const float2 *chain; size_t chain_n;
  int r = prepare_chain(h, h->d_iq, nsamples, h->own_stream, &chain, &chain_n); if (r) return r;
  if (chain_n < (size_t)(2 * h->d.N + h->d.cp + 16)) return fail(DVBT_ERR_INVALID, "segment shorter than one acquisition window");
  return segment_periods(h, chain, chain_n, h->own_stream, rep);
}

// the same on a segment that is already in device memory; synchronous (it reads the acquisition's outcome back between lock periods)
extern "C" int dvbt_rx_segment_run_device(dvbt_rx *h, const void *iq_device, size_t nsamples, void *stream, dvbt_rx_report *rep)
{
  if (!h || !iq_device) return fail(DVBT_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->prm.device));
  hipStream_t s = stream ? (hipStream_t)stream : h->own_stream;
  const float2 *chain; size_t chain_n;
  int r = prepare_chain(h, (const float2 *)iq_device, nsamples, s, &chain, &chain_n); if (r) return r;
  if (chain_n < (size_t)(2 * h->d.N + h->d.cp + 16)) return fail(DVBT_ERR_INVALID, "segment shorter than one acquisition window");
  h->cur_stream = s;
  return segment_periods(h, chain, chain_n, s, rep);
}

static int tap_info(dvbt_rx *h, int tap, void **ptr, size_t *bytes)
{
  if (!h->have_last) return fail(DVBT_ERR_STATE, "no finished segment");
  const dvbt_rx_report &r = h->last; const Dims &d = h->d;
  size_t ns = r.n_symbols > 0 ? (size_t)r.n_symbols : 0, no = r.n_out_symbols > 0 ? (size_t)r.n_out_symbols : 0;
  size_t fo = r.first_out_symbol > 0 ? (size_t)r.first_out_symbol : 0;
  switch (tap) {
    case DVBT_TAP_ACQ: *ptr = h->acq_tap; *bytes = ns * d.N * 8; break;
    case DVBT_TAP_FFT: *ptr = h->fft_out; *bytes = ns * d.N * 8; break;
    case DVBT_TAP_EQ: *ptr = h->eq ? (void *)(h->eq + fo * d.payload) : nullptr; *bytes = no * d.payload * 8; break;
    case DVBT_TAP_DEMAP: *ptr = h->labels ? (void *)(h->labels + fo * d.payload) : nullptr; *bytes = no * d.payload; break;
    case DVBT_TAP_SYMDEINT: *ptr = h->symdeint_tap; *bytes = no * d.payload; break;
    case DVBT_TAP_BITDEINT: *ptr = h->bitdeint; *bytes = no * d.payload; break;
    case DVBT_TAP_VITERBI: *ptr = h->vit; *bytes = (size_t)r.n_viterbi_bytes; break;
    case DVBT_TAP_DEINT: *ptr = h->deint_tap; *bytes = (size_t)(r.n_rs_bytes / 188) * 204; break;
    case DVBT_TAP_RS: *ptr = h->rs_out; *bytes = (size_t)r.n_rs_bytes; break;
    case DVBT_TAP_TS: *ptr = h->ts_out; *bytes = (size_t)r.n_ts_bytes; break;
    case DVBT_TAP_SYMBOL_INDEX: *ptr = h->sym_index; *bytes = (ns > 0 ? ns - 1 : 0) * 4; break;
    case DVBT_TAP_BITDEINT_LOG: *ptr = h->bd_log; *bytes = h->plog_bytes; break;
    case DVBT_TAP_BITDEINT_LP: *ptr = h->bitdeint_lp; *bytes = no * d.payload; break;
    case DVBT_TAP_SOFT: *ptr = h->soft_a; *bytes = no * d.payload * d.m; break;
    case DVBT_TAP_CSI: *ptr = h->csi ? (void *)(h->csi + fo * d.payload) : nullptr; *bytes = no * d.payload * 4; break;
    default: return fail(DVBT_ERR_INVALID, "unknown tap");
  }
  return DVBT_OK;
}

extern "C" int64_t dvbt_rx_read_tap(dvbt_rx *h, int tap, void *dst, size_t cap)
{
  if (!h || !dst) return fail(DVBT_ERR_INVALID, "null argument");
  if (tap == DVBT_TAP_CP_START) {
    size_t ns = h->have_last && h->last.n_symbols > 0 ? (size_t)h->last.n_symbols : 0;
    std::vector<SymMeta> m(ns);
    if (ns) HIPCHK(hipMemcpy(m.data(), h->meta, ns * sizeof(SymMeta), hipMemcpyDeviceToHost));
    size_t n = ns * 4 <= cap ? ns : cap / 4;
    for (size_t i = 0; i < n; i++) ((int32_t *)dst)[i] = m[i].cp_start;
    return (int64_t)(n * 4);
  }
  if (tap == DVBT_TAP_FREQ_OFFSET) {   // d_freq_offset of every demodulated symbol (reference_signals_impl.cc:715-744), out of the per-symbol sideband
    size_t ns = h->have_last && h->last.n_symbols > 1 ? (size_t)h->last.n_symbols - 1 : 0;
    std::vector<SymInfo> m(ns);
    if (ns) HIPCHK(hipMemcpy(m.data(), h->info, ns * sizeof(SymInfo), hipMemcpyDeviceToHost));
    size_t n = ns * 4 <= cap ? ns : cap / 4;
    for (size_t i = 0; i < n; i++) ((int32_t *)dst)[i] = m[i].freq_offset;
    return (int64_t)(n * 4);
  }
  void *p = nullptr; size_t bytes = 0;
  int r = tap_info(h, tap, &p, &bytes); if (r) return r;
  if (!p) return fail(DVBT_ERR_STATE, "tap not enabled (call dvbt_rx_enable_taps before the segment)");
  if (bytes > cap) bytes = cap;
  if (bytes) HIPCHK(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
  return (int64_t)bytes;
}

extern "C" void *dvbt_rx_tap_device_ptr(dvbt_rx *h, int tap)
{
  if (!h) return nullptr;
  switch (tap) {
    case DVBT_TAP_RS: return h->rs_out; case DVBT_TAP_TS: return h->ts_out; case DVBT_TAP_VITERBI: return h->vit;
    case DVBT_TAP_FFT: return h->fft_out; case DVBT_TAP_EQ: return h->eq; case DVBT_TAP_BITDEINT: return h->bitdeint;
    default: return nullptr;
  }
}

// What the proof and repair passes behind the handle's LAST launch of the Viterbi decoder did (call it behind dvbt_rx_segment_finish / a synchronous run)
extern "C" int dvbt_rx_viterbi_proof(dvbt_rx *h, dvbt_viterbi_proof *out)
{
  if (!h || !out) return fail(DVBT_ERR_INVALID, "null argument");
  if (!h->vproof.ctl) return fail(DVBT_ERR_STATE, "dvbt_rx_viterbi_proof: the handle runs the plain chunk decoders (viterbi_verify = -1, or soft decisions)");
  if (h->pending) return fail(DVBT_ERR_STATE, "dvbt_rx_viterbi_proof: a segment is in flight (dvbt_rx_segment_finish first)");
  HIPCHK(hipSetDevice(h->prm.device));
  int r[V3_CTL_HDR]; memset(r, 0, sizeof r); r[V3_CTL_UNPROVEN] = -1;
  if (h->vit_checked) HIPCHK(hipMemcpy(r, h->vproof.ctl, sizeof r, hipMemcpyDeviceToHost));
  out->chunks = r[V3_CTL_CHUNKS]; out->decoded_again = r[V3_CTL_MISMATCH]; out->sequential = r[V3_CTL_SEQ]; out->not_proven = r[V3_CTL_UNPROVEN];
  return DVBT_OK;
}
// ... and summed over every launch of the handle's decoder since it was created (the lock periods of a synchronous run are launches of their own); not_proven = -1
extern "C" int dvbt_rx_viterbi_proof_total(dvbt_rx *h, dvbt_viterbi_proof *out)
{
  if (!h || !out) return fail(DVBT_ERR_INVALID, "null argument");
  if (!h->vproof.ctl) return fail(DVBT_ERR_STATE, "dvbt_rx_viterbi_proof_total: the handle runs the plain chunk decoders (viterbi_verify = -1, or soft decisions)");
  if (h->pending) return fail(DVBT_ERR_STATE, "dvbt_rx_viterbi_proof_total: a segment is in flight (dvbt_rx_segment_finish first)");
  HIPCHK(hipSetDevice(h->prm.device));
  int acc[3] = {0, 0, 0};
  HIPCHK(hipMemcpy(acc, h->vproof.ctl + V3_CTL_ACC, sizeof acc, hipMemcpyDeviceToHost));
  out->chunks = acc[0]; out->decoded_again = acc[1]; out->sequential = acc[2]; out->not_proven = -1;
  return DVBT_OK;
}
// (chunks, chunks that are not proven when the launch ends) -- needs the final check, viterbi_verify >= 1
extern "C" int dvbt_rx_viterbi_check(dvbt_rx *h, int64_t *chunks, int64_t *unproven)
{
  if (!h || !chunks || !unproven) return fail(DVBT_ERR_INVALID, "null argument");
  if (h->prm.viterbi_verify < VIT_REPAIR_COUNT) return fail(DVBT_ERR_STATE, "dvbt_rx_viterbi_check: the handle was created without the final check (viterbi_verify >= 1)");
  dvbt_viterbi_proof p; int r = dvbt_rx_viterbi_proof(h, &p); if (r) return r;
  *chunks = p.chunks; *unproven = p.not_proven < 0 ? 0 : p.not_proven;
  return DVBT_OK;
}

extern "C" int dvbt_rx_period_taps(dvbt_rx *h, dvbt_period_tap *out, int cap)
{
  if (!h) return fail(DVBT_ERR_INVALID, "null handle");
  if (!h->bd_log) return fail(DVBT_ERR_STATE, "dvbt_rx_period_taps: dvbt_rx_enable_taps(h, 2) before the segment");
  int n = 0;
  for (const dvbt_period_tap &e : h->plog) if (e.bitdeint_bytes > 0) { if (out && n < cap) out[n] = e; n++; }
  return n;
}

extern "C" int dvbt_rx_lock_periods(dvbt_rx *h, dvbt_lock_period *out, int cap)
{
  if (!h) return fail(DVBT_ERR_INVALID, "null handle");
  const int n = (int)h->periods.size();
  for (int i = 0; i < n && i < cap && out; i++) out[i] = h->periods[i];
  return n;
}

extern "C" int dvbt_rx_walk_stats(dvbt_rx *h, dvbt_walk_stats *out)
{
  if (!h || !out) return fail(DVBT_ERR_INVALID, "null argument");
  out->small_passes = h->n_small; out->general_passes = h->n_general; out->small_chunk_calls = acq_small_cpc(h->d.cp); out->small_max_calls = ACQ_SMALL_MAX_CALLS;
  return DVBT_OK;
}

extern "C" double dvbt_rx_stage_ms(dvbt_rx *h, const char *stage)
{
  if (!h || !stage || h->n_timed == 0) return -1.0;
  if (!strcmp(stage, "total")) return h->timing == 2 ? -1.0 : h->acc_ms[ST_END] / h->n_timed;
  for (int i = 0; i < ST_END; i++) if (!strcmp(stage, kStageNames[i])) return (h->timing == 2 && i != ST_VIT) ? -1.0 : h->acc_ms[i] / h->n_timed;
  return -1.0;
}

extern "C" void dvbt_rx_destroy(dvbt_rx *h) { if (h) rx_free(h); }

// test hook: the trackers' two peak detectors (peak_detect, the reference's state machine sample by sample; peak_detect16_wave, the wavefront-wide form the
// sequential trackers use) on n cases of 16 metric values and a carried average each.  out[4k..]: npk, pos of the first, npk, pos of the second; avg_out[2k..]: d_avg after
extern "C" int dvbt_debug_peak_detect(const float *lam_host, const float *avg_host, int n, int32_t *out_host, float *avg_out_host)
{
  if (!lam_host || !avg_host || !out_host || !avg_out_host || n <= 0) return fail(DVBT_ERR_INVALID, "null argument");
  int r = need_device(); if (r) return r;
  float *dl = nullptr, *da = nullptr, *dao = nullptr; int *dout = nullptr;
  HIPCHK(hipMalloc((void **)&dl, sizeof(float) * 16 * n)); HIPCHK(hipMalloc((void **)&da, sizeof(float) * n)); HIPCHK(hipMalloc((void **)&dao, sizeof(float) * 2 * n)); HIPCHK(hipMalloc((void **)&dout, sizeof(int) * 4 * n));
  HIPCHK(hipMemcpy(dl, lam_host, sizeof(float) * 16 * n, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(da, avg_host, sizeof(float) * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(peak_selftest_kernel, dim3(n), dim3(64), 0, nullptr, (const float *)dl, (const float *)da, n, dout, dao);
  HIPCHK(hipMemcpy(out_host, dout, sizeof(int) * 4 * n, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(avg_out_host, dao, sizeof(float) * 2 * n, hipMemcpyDeviceToHost));
  (void)hipFree(dl); (void)hipFree(da); (void)hipFree(dao); (void)hipFree(dout);
  return DVBT_OK;
}

#include "dvbt_stream.inc"
#include "dvbt_rccl.inc"
#include "dvbt_blocks.inc"
