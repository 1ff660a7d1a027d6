// Spherical convolution, "windowed" kernels for gfx950 (MI355X).
//
// The general kernels of sphere_conv.hip gather the four bilinear corners of every (channel, tap, pixel) sample straight from
// global memory: 4 dependent loads and ~50 VALU instructions per sample, which bound them at 22-44 % of the fp32 MFMA peak.
// The sampling tables of the network are not arbitrary, though: a gnomonic kernel on an equirectangular grid is
// shift-invariant along the longitude axis, so all samples of a small block of output pixels fall into a compact window of
// the input (a few columns wide; a few rows taller than the block, except next to the poles where it wraps around the whole
// longitude axis).  These kernels stage that window of the input ONCE per channel chunk in LDS with plain row loads and
// build the MFMA B operand on the fly from it: 4 LDS reads + 4 FMAs per sample, no column buffer at all.
//
// Nothing is assumed about the table: the host plans every tile from the actual table values (mode_sphere_plan_build,
// same record arithmetic as the kernel, sphere_tap.h) and classifies it by the window it needs.  A table whose tiles do not
// fit any window class is reported as such and the caller uses the general kernels.
//
// Tile = 32 rows x 4 columns of output pixels (stride 1): wave v owns column w0 + v and the 32 rows h0 .. h0+31 as the N
// dimension of v_mfma_f32_32x32x2_f32 (lanes run along h: for the Cassini layout of the network that is the longitude axis,
// so the 32 lanes of a half-wave read 32 consecutive LDS words), and all (<= 128) output channels of its z-slice as 4 M-tiles.
// K is ordered chunk (8 input channels) > tap > channel pair; the two half-waves supply the two channels of a pair.
//
// Reference: sphere_conv_cuda_kernel.cu:83-113, 195-262 (im2col gather) + sphere_conv_cuda.cpp:177-205 (addmm_).
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.h"
#include "bn_internal.h"
#include "sphere_internal.h"
#include "sphere_tap.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

constexpr int RB = 2;      // MFMA N-tiles (32 rows) per tile column
constexpr int TH = 32 * RB;  // tile rows
constexpr int TW = 4;      // tile columns
constexpr int WC = 8;      // window columns
constexpr int CCH = 8;     // input channels per chunk
constexpr int KT = 9;      // taps (3x3 kernels)
constexpr int MTW = 4;     // M-tiles (32 output channels each) per wave
constexpr int NTHREADS = 64 * TW * RB;  // one wave per (column, 32-row block)
constexpr int SROWS = NTHREADS / WC;    // window rows staged per pass
constexpr int WR_SMALL = TH + 17, WR_MID = TH + 81;  // window rows of the two compact classes (odd: conflict-free column pitch)
constexpr int WR_PIPE_MAX = 5 * SROWS;  // tallest window whose next chunk still fits in registers while the current one computes

struct WinDims {
  int B, Ci, H, W, Co, G;
  int Cig, Cog;
  int NCH;  // channel chunks
  int MG;   // 128-channel output slices per group
  int wr;   // window rows of the wrap-around class (H + 1)
  int wrap_pipe;  // wrap-around class double-buffered (fits LDS and registers) or staged in place
  int sh, sw;     // element strides of h and w in the activation tensors: (W, 1) = NCHW; (1, H) = planes stored transposed
  int accumulate;
};

__host__ __device__ constexpr int chan_pitch(int wr) {
  // 8 columns of wr rows, padded so that consecutive channels are 32 banks apart (the two half-waves of a B read)
  return WC * wr + ((32 - (WC * wr) % 64) + 64) % 64;
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// wp[(((g*MG + mg)*NCH + ch)*KT + tap)*MTW + m][lane] (float4 = 4 k-steps) = W[g*Cog + mg*128 + m*32 + (lane&31)][ch*8 + 2*s + (lane>>5)][tap]
//   fold != 0: output channel co is scaled by the folded BatchNorm scale of `bn`; block 0 writes the shifts to wp[total + channel]
__global__ void pack_w_win(const float* __restrict__ w, float* __restrict__ wp, WinDims d, int fold, mode_bn_epilogue bn) {
  const long long total = (long long)d.G * d.MG * d.NCH * KT * MTW * 64 * 4;
  if (fold && blockIdx.x == 0)
    for (int o = threadIdx.x; o < d.Co; o += blockDim.x) wp[total + o] = fold_shift(bn, o);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int s = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    long long r = idx >> 8;
    const int m = (int)(r % MTW);
    r /= MTW;
    const int tap = (int)(r % KT);
    r /= KT;
    const int ch = (int)(r % d.NCH);
    r /= d.NCH;
    const int mg = (int)(r % d.MG);
    const int g = (int)(r / d.MG);
    const int co = mg * 128 + m * 32 + (lane & 31);
    const int c = ch * CCH + 2 * s + (lane >> 5);
    float v = 0.f;
    if (co < d.Cog && c < d.Cig) {
      v = w[((long long)(g * d.Cog + co) * d.Cig + c) * KT + tap];
      if (fold) v *= fold_scale(bn, g * d.Cog + co);
    }
    wp[idx] = v;
  }
}

// One tile.  WR_T > 0: compile-time window rows; WR_T == 0: window rows = d.wr (wrap-around class).  PIPE: double-buffered,
// the next chunk's rows are prefetched into registers under the MFMA phase, in NPH phases of CCH/NPH channels each (a tall
// window has too many rows to hold a whole chunk in registers); otherwise single buffer, staged in place.
template <int WR_T, bool PIPE, int NRB, int NPH, bool EPI>
__device__ __forceinline__ void fwd_tile(const float* __restrict__ x, const float* __restrict__ pos, const float4* __restrict__ wp,
                                         float* __restrict__ y, const WinDims& d, int h0, int w0, int rbase, int cbase, float* smem,
                                         const Epi& epi) {
  const int WRP = WR_T > 0 ? WR_T : d.wr;
  const int CP = chan_pitch(WRP);
  const int bufsz = CCH * CP;

  const int b = blockIdx.y;
  const int g = blockIdx.z / d.MG, mg = blockIdx.z % d.MG;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5;
  const int h = h0 + (wave / TW) * 32 + (lane & 31), w = w0 + (wave % TW);
  const bool pix_ok = h < d.H && w < d.W;
  const long long HW = (long long)d.H * d.W;

  // sampling records of this lane's pixel: window offset of the first corner + 4 weights per tap, kept in registers
  int roff[KT];
  float4 rw[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    int r0 = 0, c0 = 0;
    float4 wt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pix_ok) {
      const long long idx = (long long)h * d.W + w;
      mode::tap_record_fixed(pos[(2 * k) * HW + idx], pos[(2 * k + 1) * HW + idx], d.H, d.W, r0, c0, wt);
    }
    int lr = r0 - rbase;
    if (lr < 0) lr += d.H;
    const int lc = c0 - cbase;
    const bool dead = wt.x == 0.f && wt.y == 0.f && wt.z == 0.f && wt.w == 0.f;
    roff[k] = (dead || !pix_ok) ? 0 : lc * WRP + lr;
    rw[k] = wt;
  }

  // every LDS word that can be read must be finite: zero everything once (corners with weight 0 may point anywhere inside)
  const int lds_floats = (PIPE ? 2 : 1) * bufsz + WRP + 8;
  for (int i = tid; i < lds_floats; i += NTHREADS) smem[i] = 0.f;
  __syncthreads();

  // staging: thread -> (window column, row within a pass of SROWS rows)
  // lanes run along the contiguous axis of the planes: columns for NCHW, rows when the planes are stored transposed
  const int scol = d.sh == 1 ? tid / SROWS : tid & (WC - 1), srow = d.sh == 1 ? tid % SROWS : tid / WC;
  const int gcol = cbase + scol;
  const bool col_ok = gcol < d.W;
  const float* xg = x + ((long long)b * d.Ci + (long long)g * d.Cig) * HW + (col_ok ? gcol * d.sw : 0);
  int rowoff[NRB];  // global offset of the source row of each pass (rows wrap around the axis; H*W < 2^30)
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) {
    const int r = rb * SROWS + srow;
    rowoff[rb] = ((rbase + (r < WRP ? r : 0)) % d.H) * d.sh;
  }
  constexpr int CPH = CCH / NPH;  // channels per staging phase
  float lv[PIPE ? CPH * NRB : 1];

  auto issue = [&](int ch, int ph) {
#pragma unroll
    for (int cc = 0; cc < CPH; ++cc) {
      const int chan = ch * CCH + ph * CPH + cc;
      const float* xc = xg + (long long)(chan < d.Cig ? chan : 0) * HW;
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) lv[cc * NRB + rb] = xc[rowoff[rb]];
    }
  };
  auto commit = [&](int ch, int ph, float* buf) {
#pragma unroll
    for (int cc = 0; cc < CPH; ++cc) {
      const int c = ph * CPH + cc;
      const bool ok = col_ok && (ch * CCH + c < d.Cig);
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        const int r = rb * SROWS + srow;
        if (r < WRP) buf[c * CP + scol * WRP + r] = ok ? lv[cc * NRB + rb] : 0.f;
      }
    }
  };
  auto stage_now = [&](int ch, float* buf) {  // tall windows: plain loop, 4 loads in flight per thread
    for (int r = srow; r < WRP; r += SROWS) {
      const int ro = ((rbase + r) % d.H) * d.sh;
#pragma unroll
      for (int c4 = 0; c4 < CCH; c4 += 4) {
        float t4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int chan = ch * CCH + c4 + c;
          t4[c] = xg[(long long)(chan < d.Cig ? chan : 0) * HW + ro];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) buf[(c4 + c) * CP + scol * WRP + r] = (col_ok && ch * CCH + c4 + c < d.Cig) ? t4[c] : 0.f;
      }
    }
  };

  f32x16 acc[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  const float4* wpa = wp + ((long long)(g * d.MG + mg) * d.NCH) * KT * MTW * 64 + lane;
  const int nsteps = d.NCH * KT;
  float4 a_cur[MTW], a_nxt[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) a_cur[m] = wpa[m * 64];

  // B operand of tap k for the 4 channel pairs of the chunk in `buf`: 4 LDS reads (load_raw) + 4 FMAs (combine, operand
  // order of cu:111) each.  The two halves are separate so that the reads of the NEXT tap can be issued before the MFMAs of
  // the current one and combined after them; the scheduling barriers keep the compiler from sinking the prefetches (weights
  // from L2, window words from LDS) down to their first use, which would expose their latency on every tap.
  auto load_raw = [&](const float* buf, int k, float (&raw)[16]) {
    const float* p = buf + half * CP + roff[k];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float* q = p + 2 * s * CP;
      raw[s * 4 + 0] = q[0];
      raw[s * 4 + 1] = q[WRP];
      raw[s * 4 + 2] = q[1];
      raw[s * 4 + 3] = q[WRP + 1];
    }
  };
  auto combine = [&](const float (&raw)[16], int k, float (&v)[4]) {
    const float4 tw = rw[k];
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = tw.x * raw[s * 4] + tw.y * raw[s * 4 + 1] + tw.z * raw[s * 4 + 2] + tw.w * raw[s * 4 + 3];
  };

  if (PIPE) {
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      issue(0, ph);
      commit(0, ph, smem);
    }
    __syncthreads();
  }
  constexpr int TPP = KT / NPH;  // taps per staging phase (the last phase also takes the remainder)
  for (int ch = 0; ch < d.NCH; ++ch) {
    float* cur = smem + (PIPE ? (ch & 1) * bufsz : 0);
    float* nxt = smem + (PIPE ? ((ch + 1) & 1) * bufsz : 0);
    const bool more = ch + 1 < d.NCH;
    if (!PIPE) {
      if (ch > 0) __syncthreads();  // everyone is done reading the previous chunk
      stage_now(ch, cur);
      __syncthreads();
    }
    float vb[4], raw[16];
    load_raw(cur, 0, raw);
    combine(raw, 0, vb);
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const int step = ch * KT + k;
      const int nstep = step + 1 < nsteps ? step + 1 : step;
      if (PIPE && k % TPP == 0 && k / TPP < NPH && more) issue(ch + 1, k / TPP);  // rows of the next chunk fly under the MFMAs below
#pragma unroll
      for (int m = 0; m < MTW; ++m) a_nxt[m] = wpa[((long long)nstep * MTW + m) * 64];
      if (k + 1 < KT) load_raw(cur, k + 1, raw);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float a0 = s == 0 ? a_cur[0].x : s == 1 ? a_cur[0].y : s == 2 ? a_cur[0].z : a_cur[0].w;
        const float a1 = s == 0 ? a_cur[1].x : s == 1 ? a_cur[1].y : s == 2 ? a_cur[1].z : a_cur[1].w;
        const float a2 = s == 0 ? a_cur[2].x : s == 1 ? a_cur[2].y : s == 2 ? a_cur[2].z : a_cur[2].w;
        const float a3 = s == 0 ? a_cur[3].x : s == 1 ? a_cur[3].y : s == 2 ? a_cur[3].z : a_cur[3].w;
        acc[0] = mfma32(a0, vb[s], acc[0]);
        acc[1] = mfma32(a1, vb[s], acc[1]);
        acc[2] = mfma32(a2, vb[s], acc[2]);
        acc[3] = mfma32(a3, vb[s], acc[3]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (k + 1 < KT) combine(raw, k + 1, vb);
#pragma unroll
      for (int m = 0; m < MTW; ++m) a_cur[m] = a_nxt[m];
      if (PIPE && more && (k == KT - 1 || (k % TPP == TPP - 1 && k / TPP < NPH - 1)))
        commit(ch + 1, k == KT - 1 ? NPH - 1 : k / TPP, nxt);  // the other buffer: nobody reads it now
    }
    if (PIPE) __syncthreads();
  }

  if (pix_ok) {
    float* yb = y + ((long long)b * d.Co + (long long)g * d.Cog + (long long)mg * 128) * HW + (long long)h * d.sh + (long long)w * d.sw;
    const int cmax = d.Cog - mg * 128;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      // eval mode: the shifts and the residual of this output tile, requested together ahead of its stores (see sphere_fwd_split_kernel)
      float res[16], shv[16];
      if (EPI) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = min(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, cmax - 1);
          shv[r] = epi.shift[g * d.Cog + mg * 128 + co];
          res[r] = 0.f;
        }
        if (epi.add) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = min(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, cmax - 1);
            res[r] = epi.add[(yb - y) + (long long)co * HW];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      auto emit = [&](int r) {
        const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (EPI) {  // folded BatchNorm shift (+ residual) (+ ReLU) on the way out
          const float v = (acc[m][r] + shv[r]) + res[r];
          yb[(long long)co * HW] = epi.relu ? relu_nan(v) : v;
        } else {
          yb[(long long)co * HW] = acc[m][r];
        }
      };
      if (m * 32 + 32 <= cmax) {
#pragma unroll
        for (int r = 0; r < 16; ++r) emit(r);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < cmax) emit(r);
      }
      if (EPI) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// grid = (tiles, B, G*MG); tiles[i] = (h0, w0, rbase, cbase | class << 16).  All window classes run in ONE launch so that
// the few tall-window tiles next to the poles overlap with the rest instead of forming an under-filled tail of their own.
template <bool EPI>
__global__ __launch_bounds__(NTHREADS) void sphere_fwd_win_kernel(const float* __restrict__ x, const float* __restrict__ pos,
                                                                   const float4* __restrict__ wp, float* __restrict__ y, WinDims d,
                                                                   const int4* __restrict__ tiles, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int4 t = tiles[blockIdx.x];  // (list order = tall-window tiles first; an XCD-contiguous order would pile them on one XCD)
  const int cls = t.w >> 16, cbase = t.w & 0xffff;
  if (cls == 0)
    fwd_tile<WR_SMALL, true, (WR_SMALL + SROWS - 1) / SROWS, 1, EPI>(x, pos, wp, y, d, t.x, t.y, t.z, cbase, smem, epi);
  else if (cls == 1)
    fwd_tile<WR_MID, true, (WR_MID + SROWS - 1) / SROWS, 2, EPI>(x, pos, wp, y, d, t.x, t.y, t.z, cbase, smem, epi);
  else
    fwd_tile<0, true, WR_PIPE_MAX / SROWS, 4, EPI>(x, pos, wp, y, d, t.x, t.y, t.z, cbase, smem, epi);
}

// Wrap-around tiles of images too tall for the double-buffered form (H + 1 > 320 rows, or more LDS than there is): single
// buffer, staged in place.  A kernel of its own so that its register needs do not weigh on the main one.
template <bool EPI>
__global__ __launch_bounds__(NTHREADS) void sphere_fwd_win_tall_kernel(const float* __restrict__ x, const float* __restrict__ pos,
                                                                        const float4* __restrict__ wp, float* __restrict__ y,
                                                                        WinDims d, const int4* __restrict__ tiles, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int4 t = tiles[blockIdx.x];  // (list order = tall-window tiles first; an XCD-contiguous order would pile them on one XCD)
  fwd_tile<0, false, 1, 1, EPI>(x, pos, wp, y, d, t.x, t.y, t.z, t.w & 0xffff, smem, epi);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward w.r.t. the weight on the compact-window tiles:  gW[o][c][k] = sum_p gy[o][p] * col[c][k][p]
//   D[i = o][j = c] per tap, K = the pixels of the tile; A = gy (LDS, [o][pixel]), B = col built on the fly from the x window
//   of 32 input channels (LDS, [c][col][row], odd channel pitch: the 32 lanes of a half-wave read 32 channels).
// Work item = half a plan tile (32 rows x 4 columns) of one sample, processed column by column (32 pixels = 16 k-steps).
// grid = (S, ceil(Cig/32), G*MG); split-K slice s owns the items s, s+S, ...; 8 waves: wave v owns tap v for all 4 o-tiles,
// tap 8 is shared: wave v takes the pixels p = v (mod 8) of every column (the 8 shares are summed through LDS in wave order
// at the end).  Partial sums go to part[s][z][cg][tap][128 o][32 c], summed in fixed order by reduce_gw_win.
// The contraction runs on v_mfma_f32_32x32x1_2b_f32: ONE pixel per instruction, two o-tiles (the two 32x32 blocks) at a time.
// With one pixel per step the sampling record is the same for the whole wave, so it lives in SGPRs (scalar loads from the
// record table) and the bilinear combine is 4 VALU instructions with scalar weights -- the 32x32x2 form (two pixels per step,
// one per half-wave) needed ~14 VALU + 7 SALU instructions per MFMA for the per-half selects and measured 30 % MFMA busy.
// The sampling records (window offset + 4 weights per tap and pixel) come from a table the host builds with the plan
// (mode_sphere_plan_records): they depend on the table and the tile only, not on the sample, layer or channel group.
// Everything the next column needs (gy, records; the x window before a new item) is fetched into registers before the MFMAs
// of the current column and written to the other LDS buffer after them.
constexpr int BW_CG = 32;
constexpr int BW_TH = 32;                  // rows per work item
constexpr int BW_WR = BW_TH + 17;          // its window rows (49)
constexpr int BW_CP = WC * BW_WR + 1;      // odd channel pitch of the x window
constexpr int BW_GP = BW_TH + 1;           // gy row pitch
constexpr int BW_SLOTS = KT;                // one partial per tap (the waves' shares of tap 8 are summed in the kernel)
constexpr int BW_XW = BW_CG * BW_CP + BW_WR + 8;  // + slack: zero-weight corners may point just past the last window
constexpr int BW_COLBUF = 128 * BW_GP;  // gy column
constexpr int BW_LDS_FLOATS = 2 * BW_XW + 2 * BW_COLBUF;  // x window and gy column, both double-buffered
constexpr int BW_NXW = BW_CG / TW;  // x-window words per thread and column step: the next item's window arrives in 4 parts
constexpr int BW_NREC = KT * BW_TH;                                      // records per column (288)

// The 72 MFMAs of one column (32 pixels) of a work item for this wave: own tap (all 32 pixels, software-pipelined: while the
// MFMAs of pixel px run, the B value of pixel px+1 is combined and the LDS words of pixel px+2 are requested), then this wave's
// share of tap 8.  cb = gy column [128][BW_GP], xw = x window(s) [32][CP], rw/ro (rw8/ro8) = lane l holds the record of pixel
// l & 31 of this wave's tap (of tap 8); WRP = distance between the two window columns of a record.
template <int WRP, int CP>
__device__ __forceinline__ void bww_column(const float* cb, const float* xw, const float4 rw, const int ro, const float4 rw8,
                                           const int ro8, f32x32 (&accp)[2], f32x32 (&acc8p)[2], int wave_u, int j, int half) {
  // A operand: block 0 (lanes 0-31) = o-tile 0 / 2, block 1 (lanes 32-63) = o-tile 1 / 3
  const float* ap = cb + (half * 32 + j) * BW_GP;
  const float* xb = xw + j * CP;
  auto load_x = [&](int o, float (&raw)[4]) {  // o is wave-uniform
    const float* p = xb + o;
    raw[0] = p[0];
    raw[1] = p[WRP];
    raw[2] = p[1];
    raw[3] = p[WRP + 1];
  };
  auto bcast = [&](float v, int px) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), px)); };
  auto combine = [&](const float4& w, int px, const float (&raw)[4]) {  // one fma chain: keeps the compiler from SLP-packing
    return __builtin_fmaf(bcast(w.w, px), raw[3],
                          __builtin_fmaf(bcast(w.z, px), raw[2], __builtin_fmaf(bcast(w.y, px), raw[1], bcast(w.x, px) * raw[0])));
  };
  // own tap: all 32 pixels, software-pipelined three deep: while the MFMAs of pixel px run, the B value of pixel px+1 is
  // combined (4 FMAs on words read one step earlier) and the LDS words of pixel px+2 are requested
  {
    float raw[2][4], a[3][2];
#pragma unroll
    for (int p0 = 0; p0 < 2; ++p0) {
      load_x(__builtin_amdgcn_readlane(ro, p0), raw[p0]);
      a[p0][0] = ap[p0];
      a[p0][1] = ap[64 * BW_GP + p0];
    }
    float bv = combine(rw, 0, raw[0]);
#pragma unroll
    for (int px = 0; px < BW_TH; ++px) {
      const int c3 = px % 3, n3 = (px + 2) % 3;
      float bv_next = 0.f;
      if (px + 1 < BW_TH) bv_next = combine(rw, px + 1, raw[(px + 1) & 1]);
      if (px + 2 < BW_TH) {
        load_x(__builtin_amdgcn_readlane(ro, px + 2), raw[px & 1]);  // the slot of pixel px is free again
        a[n3][0] = ap[px + 2];
        a[n3][1] = ap[64 * BW_GP + px + 2];
      }
      __builtin_amdgcn_sched_barrier(0);
      accp[0] = __builtin_amdgcn_mfma_f32_32x32x1f32(a[c3][0], bv, accp[0], 0, 0, 0);
      accp[1] = __builtin_amdgcn_mfma_f32_32x32x1f32(a[c3][1], bv, accp[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      bv = bv_next;
    }
  }
  // this wave's share of tap 8: pixels wave, wave + 8, wave + 16, wave + 24
  {
    float raw[4][4], a[4][2];
#pragma unroll
    for (int i = 0; i < BW_TH / 8; ++i) {
      const int px = wave_u + 8 * i;
      load_x(__builtin_amdgcn_readlane(ro8, px), raw[i]);
      a[i][0] = ap[px];
      a[i][1] = ap[64 * BW_GP + px];
    }
#pragma unroll
    for (int i = 0; i < BW_TH / 8; ++i) {
      const float b8 = combine(rw8, wave_u + 8 * i, raw[i]);
      acc8p[0] = __builtin_amdgcn_mfma_f32_32x32x1f32(a[i][0], b8, acc8p[0], 0, 0, 0);
      acc8p[1] = __builtin_amdgcn_mfma_f32_32x32x1f32(a[i][1], b8, acc8p[1], 0, 0, 0);
    }
  }

}

// Partials of one workgroup: tap `wave` from its wave, tap 8 = the 8 waves' shares added in wave order through LDS (deterministic).
// Must be called after a barrier that ends all reads of `smem`.
__device__ __forceinline__ void bww_write_partials(float* pb, float* smem, f32x32 (&accp)[2], f32x32 (&acc8p)[2], int wave, int half,
                                                   int j, int tid) {
#pragma unroll
  for (int pr = 0; pr < 2; ++pr)
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const int o = (2 * pr + (r >> 4)) * 32 + (r & 3) + 8 * ((r & 15) >> 2) + 4 * half;
      pb[((long long)wave * 128 + o) * BW_CG + j] = accp[pr][r];
    }
  // tap 8: the 8 waves' shares, added in wave order through LDS (deterministic), then written as one partial
  float* red = smem;  // [128 o][32 c]; the item loop ended with a barrier, nobody reads the windows any more
  for (int v = 0; v < 8; ++v) {
    if (wave == v) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int r = 0; r < 32; ++r) {
          const int o = (2 * pr + (r >> 4)) * 32 + (r & 3) + 8 * ((r & 15) >> 2) + 4 * half;
          float* q = red + o * BW_CG + j;
          *q = v == 0 ? acc8p[pr][r] : *q + acc8p[pr][r];
        }
    }
    __syncthreads();
  }
  for (int i = tid; i < 128 * BW_CG; i += NTHREADS) pb[(long long)8 * 128 * BW_CG + i] = red[i];
}

__global__ __launch_bounds__(NTHREADS) void sphere_bww_win_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                   float* __restrict__ part, WinDims d, const int4* __restrict__ tiles,
                                                                   const float4* __restrict__ rec_w, const int* __restrict__ rec_off,
                                                                   int ntiles, int S) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* colbuf = smem + 2 * BW_XW;  // smem: [2][x window 32 x BW_CP] [2][gy column]
  constexpr int WRP = BW_WR;
  const int s = blockIdx.x, cg = blockIdx.y;
  const int g = blockIdx.z / d.MG, mg = blockIdx.z % d.MG;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, half = lane >> 5;
  const long long HW = (long long)d.H * d.W;
  const int T = ntiles * 2 * d.B;  // (sample, tile, half) items
  const int NCG = gridDim.y;

  f32x32 accp[2], acc8p[2];  // [o-tile pair]: elements 0-15 = first tile of the pair, 16-31 = second
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    accp[0][r] = accp[1][r] = 0.f;
    acc8p[0][r] = acc8p[1][r] = 0.f;
  }
  for (int i = tid; i < BW_LDS_FLOATS; i += NTHREADS) smem[i] = 0.f;  // every word that may be read is finite

  // prefetch registers
  float pxw[BW_NXW];  // x window of the next item
  float pgy[8];       // gy of the next column: o = (tid >> 5) + 16 u, pixel tid & 31
  float4 prw, prw8;   // records of the next column: lane l holds pixel l & 31 of this wave's tap and of tap 8
  int pro, pro8;

  const int omax = d.Cog - mg * 128;
  const int cmax = d.Cig - cg * BW_CG;
  const int gpx = tid & (BW_TH - 1), go0 = tid >> 5;

  struct Item {  // geometry of a work item, wave-uniform
    int b, ti, hf, h0, w0, rbase, cbase;
  };
  auto item_of = [&](int t) {
    Item it;
    it.b = t / (ntiles * 2);
    const int r = t - it.b * ntiles * 2;
    it.ti = r >> 1;
    it.hf = r & 1;
    const int4 tl = tiles[it.ti];
    it.h0 = tl.x + it.hf * BW_TH;
    it.w0 = tl.y;
    it.rbase = (tl.z + it.hf * BW_TH) % d.H;
    it.cbase = tl.w & 0xffff;
    return it;
  };
  // x window: thread -> (column, row < 49), one word per channel; addresses are base + channel * stride.  Lanes run along the
  // contiguous axis of the planes (columns for NCHW, rows for transposed planes).
  const int xcol = d.sh == 1 ? tid >> 6 : tid & (WC - 1), xrow = d.sh == 1 ? tid & 63 : tid >> 3;
  const bool xrow_ok = xrow < WRP;
  // Loads are issued unconditionally from clamped addresses and masked when they are written to LDS: a load whose only
  // consumer is a select on the same condition gets sunk into a branch by the compiler, with a full vmcnt(0) wait per load
  // (measured: 7.6k cycles per column spent "issuing" 12 loads).
  bool pxw_ok = false;   // row / column of this thread's window words inside the image (next item)
  unsigned pgy_ok = 0;   // bit u: pgy[u] is a real value (next column)
  auto issue_xw = [&](const Item& it, int part) {  // channels part*8 .. part*8+7 of the window of item `it`
    pxw_ok = xrow_ok && it.cbase + xcol < d.W;
    const int grow = (it.rbase + (xrow_ok ? xrow : 0)) % d.H;
    const float* xg = x + ((long long)it.b * d.Ci + (long long)g * d.Cig + (long long)cg * BW_CG) * HW +
                      (pxw_ok ? (long long)grow * d.sh + (long long)(it.cbase + xcol) * d.sw : 0);
#pragma unroll
    for (int c = 0; c < BW_NXW; ++c) pxw[c] = xg[(long long)min(part * BW_NXW + c, cmax - 1) * HW];
  };
  auto commit_xw = [&](float* xwdst, int part) {
    if (xrow_ok) {
      float* dst = xwdst + xcol * WRP + xrow;
#pragma unroll
      for (int c = 0; c < BW_NXW; ++c) dst[(part * BW_NXW + c) * BW_CP] = (pxw_ok && part * BW_NXW + c < cmax) ? pxw[c] : 0.f;
    }
  };
  auto issue_col = [&](const Item& it, int wc) {
    const int h = it.h0 + gpx, w = it.w0 + wc;
    const bool pok = h < d.H && w < d.W;
    const float* gyb = gy + ((long long)it.b * d.Co + (long long)g * d.Cog + (long long)mg * 128) * HW +
                       (pok ? (long long)h * d.sh + (long long)w * d.sw : 0);
    pgy_ok = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int o = go0 + 16 * u;
      pgy[u] = gyb[(long long)min(o, omax - 1) * HW];
      pgy_ok |= (pok && o < omax) ? (1u << u) : 0u;
    }
    const long long rcol = (((long long)it.ti * 2 + it.hf) * TW + wc) * BW_NREC;
    prw = rec_w[rcol + wave * BW_TH + j];
    pro = rec_off[rcol + wave * BW_TH + j];
    prw8 = rec_w[rcol + 8 * BW_TH + j];
    pro8 = rec_off[rcol + 8 * BW_TH + j];
  };
  auto commit_col = [&](float* cb) {
#pragma unroll
    for (int u = 0; u < 8; ++u) cb[(go0 + 16 * u) * BW_GP + gpx] = (pgy_ok >> u & 1u) ? pgy[u] : 0.f;
  };

  // first item: window and first column
  Item cur = item_of(s);
  __syncthreads();  // zero fill done
  for (int part = 0; part < TW; ++part) {
    issue_xw(cur, part);
    commit_xw(smem, part);
  }
  issue_col(cur, 0);
  commit_col(colbuf);
  __syncthreads();

  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  int buf = 0, xbuf = 0;
  for (int t = s; t < T; t += S) {
    const bool more_items = t + S < T;
    const Item nxt = item_of(more_items ? t + S : t);
    const float* xw = smem + xbuf * BW_XW;
    for (int wc = 0; wc < TW; ++wc) {
      const float* cb = colbuf + buf * BW_COLBUF;
      const bool last_col = wc == TW - 1;
      const bool have_next = !last_col || more_items;
      // The sampling record of a pixel is the same for every lane: lane l holds the record of pixel l & 31 (vector loads,
      // prefetched with the column) and each step broadcasts its pixel's record with v_readlane -- no scalar-memory loads
      // in the loop, whose out-of-order return would force a full LDS drain on every use.
      const float4 rw = prw, rw8 = prw8;
      const int ro = pro, ro8 = pro8;
      if (have_next) issue_col(last_col ? nxt : cur, last_col ? 0 : wc + 1);
      if (more_items) issue_xw(nxt, wc);  // a quarter of the next item's window per column, into the other window buffer

      bww_column<BW_WR, BW_CP>(cb, xw, rw, ro, rw8, ro8, accp, acc8p, wave_u, j, half);

      // the other buffers: nobody reads them now
      if (have_next) commit_col(colbuf + (buf ^ 1) * BW_COLBUF);
      if (more_items) commit_xw(smem + (xbuf ^ 1) * BW_XW, wc);
      __syncthreads();
      buf ^= 1;
    }
    cur = nxt;
    xbuf ^= 1;
  }

  bww_write_partials(part + ((((long long)s * gridDim.z + blockIdx.z) * NCG + cg) * BW_SLOTS) * (128 * BW_CG), smem, accp, acc8p, wave,
                     half, j, tid);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same contraction for the tiles next to the poles, where the nine taps of one output column sample row ranges that lie far
// apart (so no single compact window exists): a work item is ONE column of 32 pixels, and the x window is nine small windows,
// one per tap: [32 ch][9 taps][2 columns][34 rows] (78 KB).  Everything of an item is staged in place (no cross-item prefetch:
// these items are 12.5 % of the pixels), then the k-loop of the compact-window kernel runs unchanged on it.
// pitems[i] = (h0, w, rbase[9], cbase[9]); records [item][tap][32] hold offsets into the per-tap layout.
constexpr int BP_WR = BW_TH + 2;         // rows per tap window
constexpr int BP_TAPW = 2 * BP_WR;       // floats per tap window: [2 cols][34 rows]
constexpr int BP_CP = KT * BP_TAPW + 1;  // odd channel pitch (613)
constexpr int BP_XW = BW_CG * BP_CP + BP_WR + 8;
constexpr int BP_LDS_FLOATS = BP_XW + BW_COLBUF;
constexpr int BP_ITEM_INTS = 2 + 2 * KT;

__global__ __launch_bounds__(NTHREADS) void sphere_bww_polar_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                     float* __restrict__ part, WinDims d, const int* __restrict__ pitems,
                                                                     const float4* __restrict__ rec_w, const int* __restrict__ rec_off,
                                                                     int nitems, int S, int s_base) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xw = smem;
  float* cb = smem + BP_XW;
  const int s = blockIdx.x, cg = blockIdx.y;
  const int g = blockIdx.z / d.MG, mg = blockIdx.z % d.MG;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, half = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const long long HW = (long long)d.H * d.W;
  const int T = nitems * d.B;
  const int omax = d.Cog - mg * 128, cmax = d.Cig - cg * BW_CG;
  const int gpx = tid & (BW_TH - 1), go0 = tid >> 5;

  f32x32 accp[2], acc8p[2];
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    accp[0][r] = accp[1][r] = 0.f;
    acc8p[0][r] = acc8p[1][r] = 0.f;
  }
  for (int i = tid; i < BP_LDS_FLOATS; i += NTHREADS) smem[i] = 0.f;

  // this thread's window words: element e = tid (and tid + 512 < 612) of the [9][2][34] per-channel layout
  constexpr int NE = KT * BP_TAPW;  // 612
  const int e0 = tid, e1 = tid + NTHREADS;
  const bool has1 = e1 < NE;
  const int k0 = e0 / BP_TAPW, k1 = (has1 ? e1 : 0) / BP_TAPW;
  const int col0 = (e0 % BP_TAPW) / BP_WR, r0 = e0 % BP_WR;
  const int col1 = ((has1 ? e1 : 0) % BP_TAPW) / BP_WR, r1 = (has1 ? e1 : 0) % BP_WR;

  for (int t = s; t < T; t += S) {
    const int b = t / nitems, it = t - b * nitems;
    const int* pi = pitems + (long long)it * BP_ITEM_INTS;
    const int h0 = pi[0], w = pi[1];
    __syncthreads();  // previous item consumed (first pass: zero fill done)
    {
      const int rb0 = pi[2 + k0], cb0 = pi[2 + KT + k0], rb1 = pi[2 + k1], cb1 = pi[2 + KT + k1];
      const bool ok0 = cb0 + col0 < d.W, ok1 = has1 && cb1 + col1 < d.W;
      const float* xg = x + ((long long)b * d.Ci + (long long)g * d.Cig + (long long)cg * BW_CG) * HW;
      const long long a0 = ok0 ? (long long)((rb0 + r0) % d.H) * d.sh + (long long)(cb0 + col0) * d.sw : 0;
      const long long a1 = ok1 ? (long long)((rb1 + r1) % d.H) * d.sh + (long long)(cb1 + col1) * d.sw : 0;
#pragma unroll 1
      for (int c8 = 0; c8 < BW_CG; c8 += 8) {  // 16 unconditional loads in flight, masked at the store
        float v0[8], v1[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const long long co = (long long)min(c8 + c, cmax - 1) * HW;
          v0[c] = xg[co + a0];
          v1[c] = xg[co + a1];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          xw[(c8 + c) * BP_CP + e0] = (ok0 && c8 + c < cmax) ? v0[c] : 0.f;
          if (has1) xw[(c8 + c) * BP_CP + e1] = (ok1 && c8 + c < cmax) ? v1[c] : 0.f;
        }
      }
    }
    {
      const int h = h0 + gpx;
      const bool pok = h < d.H && w < d.W;
      const float* gyb = gy + ((long long)b * d.Co + (long long)g * d.Cog + (long long)mg * 128) * HW +
                         (pok ? (long long)h * d.sh + (long long)w * d.sw : 0);
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = gyb[(long long)min(go0 + 16 * u, omax - 1) * HW];
#pragma unroll
      for (int u = 0; u < 8; ++u) cb[(go0 + 16 * u) * BW_GP + gpx] = (pok && go0 + 16 * u < omax) ? v[u] : 0.f;
    }
    const long long rbase = (long long)it * BW_NREC;
    const float4 rw = rec_w[rbase + wave * BW_TH + j], rw8 = rec_w[rbase + 8 * BW_TH + j];
    const int ro = rec_off[rbase + wave * BW_TH + j], ro8 = rec_off[rbase + 8 * BW_TH + j];
    __syncthreads();
    bww_column<BP_WR, BP_CP>(cb, xw, rw, ro, rw8, ro8, accp, acc8p, wave_u, j, half);
  }
  __syncthreads();
  bww_write_partials(part + ((((long long)(s_base + s) * gridDim.z + blockIdx.z) * gridDim.y + cg) * BW_SLOTS) * (128 * BW_CG), smem, accp,
                     acc8p, wave, half, j, tid);
}

// gw[o][c][k] += sum over slices (fixed order) of the tap's partial; one thread per (z, cg, k, o, c), c fastest:
// coalesced reads of the partials
__global__ void reduce_gw_win(const float* __restrict__ part, float* __restrict__ gw, WinDims d, int S, int NCG) {
  const int GZ = d.G * d.MG;
  const long long total = (long long)GZ * NCG * KT * 128 * BW_CG;
  const long long stride = (long long)GZ * NCG * BW_SLOTS * 128 * BW_CG;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int cl = (int)(idx % BW_CG);
    long long r = idx / BW_CG;
    const int row = (int)(r % 128);
    r /= 128;
    const int k = (int)(r % KT);
    r /= KT;
    const int cg = (int)(r % NCG);
    const int z = (int)(r / NCG);
    const int g = z / d.MG, mg = z % d.MG;
    const int ol = mg * 128 + row, c = cg * BW_CG + cl;
    if (ol >= d.Cog || c >= d.Cig) continue;
    const float* pp = part + ((((long long)z * NCG + cg) * BW_SLOTS + k) * 128 + row) * BW_CG + cl;
    // fixed association (4 interleaved running sums over the slices): deterministic, 4 independent loads in flight
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int e = 0;
    for (; e + 3 < S; e += 4) {
      a0 += pp[(long long)e * stride];
      a1 += pp[(long long)(e + 1) * stride];
      a2 += pp[(long long)(e + 2) * stride];
      a3 += pp[(long long)(e + 3) * stride];
    }
    for (; e < S; ++e) a0 += pp[(long long)e * stride];
    const float sum = (a0 + a1) + (a2 + a3);
    gw[((long long)(g * d.Cog + ol) * d.Cig + c) * KT + k] += sum;
  }
}

// Class 0 also promises the weight-gradient kernel that each 32-row half of the tile fits a 49-row window starting at
// rbase (+32 for the second half): true for shift-invariant tables, checked for all.
bool halves_fit(const float* pos_host, int H, int W, int KK, int h0, int w0, int rbase, int cbase) {
  const long long HW = (long long)H * W;
  for (int hf = 0; hf < 2; ++hf) {
    const int rb = (rbase + hf * BW_TH) % H;
    for (int k = 0; k < KK; ++k)
      for (int h = h0 + hf * BW_TH; h < std::min(h0 + (hf + 1) * BW_TH, H); ++h)
        for (int w = w0; w < std::min(w0 + TW, W); ++w) {
          int r0, c0;
          float4 wt;
          const long long idx = (long long)h * W + w;
          if (!mode::tap_record_fixed(pos_host[(2 * k) * HW + idx], pos_host[(2 * k + 1) * HW + idx], H, W, r0, c0, wt)) continue;
          if (wt.x == 0.f && wt.y == 0.f && wt.z == 0.f && wt.w == 0.f) continue;
          const int lr = ((r0 - rb) % H + H) % H;
          if (lr + 1 >= BW_WR || c0 - cbase < 0 || c0 - cbase + 1 >= WC) return false;
        }
  }
  return true;
}

int bww_win_splits(const WinDims& d, int ntiles) {
  const int T = ntiles * 2 * d.B;
  const int wgs_per_slice = mode::cdiv(d.Cig, BW_CG) * d.G * d.MG;
  const int per = std::max(1, mode::cdiv((long long)T * wgs_per_slice, kNumCU));  // items per workgroup for ~one workgroup per CU
  return std::max(1, mode::cdiv(T, per));
}


// =====================================================================================================================
// Forward on the small-window tiles with the split-bf16 arithmetic of conv3d_split.hip: fp32 operands split exactly
// into three bf16 pieces, six v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation.
//
// Same tile (64 rows x 4 columns), window (81 rows x 8 columns per channel, fp32, double-buffered) and sampling records as
// fwd_tile; what differs is who does what.  K of an MFMA is 16 input channels of one tap, so a chunk is 16 channels deep and
//   * SAMPLING: wave v builds the B fragment of its own 32 pixels (column v % 4, row block v / 4) for tap t + 1 -- 8 channels
//     per lane (lanes 0..31 channels 0..7, lanes 32..63 channels 8..15), 4 LDS words + 4 FMAs each, then the exact 3-way split --
//     and writes the three uint4 to an operand buffer in LDS (2 x 24 KB);
//   * MATRIX: wave v owns output-channel tile v % 4 for the 4 pixel groups of its row block: per tap it reads the 4 x 3
//     fragments the waves of its row block have written and issues 24 MFMAs against ONE weight fragment set (3 KB from L2 per
//     tap and wave -- with all four output tiles per wave, as in fwd_tile, the weights would be 12 KB per tap and wave, and 96 KB per
//     tap for the workgroup through a 64 B/clk L1 lasts as long as the tap's MFMAs at this matrix rate).
// Both run in one instruction stream per wave, the sampling of tap t + 1 under the MFMAs of tap t; one LDS barrier per tap.
// The window of the next chunk is loaded under taps 0..6 and stored under taps 1..7.
constexpr int SP_CCH = 16;
constexpr int SP_CP = chan_pitch(WR_SMALL);
constexpr int SP_WIN = SP_CCH * SP_CP;                      // floats per window buffer
constexpr int SP_WIN_FLOATS = ((2 * SP_WIN + WR_SMALL + 8 + 3) / 4) * 4;  // both buffers + slack, 16-byte multiple
constexpr int SP_OP = 8 * 3 * 64;                           // uint4 per operand buffer
constexpr size_t SP_LDS_BYTES = (size_t)SP_WIN_FLOATS * sizeof(float) + 2 * (size_t)SP_OP * sizeof(uint4);
constexpr size_t SP3_LDS_BYTES = SP_LDS_BYTES + (size_t)SP_OP * sizeof(uint4);  // the forward keeps three operand buffers
static_assert(SP3_LDS_BYTES <= 160 * 1024, "windows + three operand buffers must fit the 160 KB of a CU");

typedef __bf16 sp_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t sp_pack2(float a, float b) {
  const sp_f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, sp_bf16x2));
}
__device__ __forceinline__ void sp_split2(float a, float b, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  // (the two subtractions of a pair must stay scalar: SLP-packed into v_pk_add_f32 each costs ~9 cycles of the MATRIX pipe -- packed
  // fp32 instructions do not overlap with MFMAs on gfx950, plain ones do; tools/experiments/mfma_op_cost.hip)
  p1 = sp_pack2(a, b);
  float ra = a - __builtin_bit_cast(float, p1 << 16), rb = b - __builtin_bit_cast(float, p1 & 0xffff0000u);
  asm("" : "+v"(ra), "+v"(rb));
  p2 = sp_pack2(ra, rb);
  float sa = ra - __builtin_bit_cast(float, p2 << 16), sb = rb - __builtin_bit_cast(float, p2 & 0xffff0000u);
  asm("" : "+v"(sa), "+v"(sb));
  p3 = sp_pack2(sa, sb);
}
__device__ __forceinline__ f32x16 sp_mfma(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sp_bf16x8, a), __builtin_bit_cast(sp_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void sp_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- the two-piece fp16 arithmetic of conv3d_split.hip for the small-window tiles of the TRAINING forward: two fp16 pieces
// per value (v_cvt_pk_f16_f32, round to nearest even; the remainder is exact in fp32), three v_mfma_f32_32x32x16_f16 per product
// (lo x hi, hi x hi, hi x lo).  Both operands are multiplied by a power of two that brings their tensor's largest finite magnitude (a
// device scalar from the caller) to [2^14, 2^15); the sampled operand gets it through its four bilinear weights, once per tile.
typedef _Float16 sp_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 sp_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float sp_f16_scale_of(float m) {  // as in conv3d_split.hip: m * scale in [2^14, 2^15)
  const unsigned e = min(max((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu, 64u), 254u);
  return m == 0.f ? 1.f : __builtin_bit_cast(float, (268u - e) << 23);
}
__device__ __forceinline__ void sp_split2_f16(float a, float b, uint32_t& p1, uint32_t& p2) {
  const sp_f32x2 v = {a, b};
  const sp_f16x2 h1 = __builtin_convertvector(v, sp_f16x2);
  p1 = __builtin_bit_cast(uint32_t, h1);
  // (round 6) the remainders a - (float)h as ONE instruction each: v_fma_mix_f32 reads the fp16 half in place (1.0 * a - h, the same
  // exact difference).  Written as `a - (float)h` it is a v_cvt_f32_f16 + a v_sub_f32 per value -- 40 / 72 / 104 of the 445 / 970 / 1 637
  // vector instructions of the forward / input-gradient / weight-gradient loops -- and fma(-1, h, a) is folded back to that; the asm
  // statement also does what the empty one before it did: it keeps the pair's two chains scalar (see sp_split2).
  float ra, rb;
  asm("v_fma_mix_f32 %0, 1.0, %3, -%2 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mix_f32 %1, 1.0, %4, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(ra), "=&v"(rb)
      : "v"(p1), "v"(a), "v"(b));
  const sp_f32x2 r = {ra, rb};
  p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, sp_f16x2));
}
__device__ __forceinline__ f32x16 sp_mfma_f16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sp_f16x8, a), __builtin_bit_cast(sp_f16x8, b), c, 0, 0, 0);
}

// wps[(((((g*MG + mg)*NCH16 + ch)*KT + tap)*MTW + m)*3 + piece)*64 + lane] = 8 bf16: piece of W[g*Cog + mg*128 + m*32 + (lane&31)]
// [ch*16 + 8*(lane>>5) + j][tap], j = 0..7 (scaled by the folded BatchNorm scale when fold != 0; the shifts are the ones pack_w_win wrote)
// F16: two fp16 pieces of w * (the weight tensor's power-of-two scale, from amax_w[0]) in the places of pieces 0 and 1 (no fold)
template <bool F16>
__global__ void pack_w_win_split(const float* __restrict__ w, uint4* __restrict__ wps, WinDims d, int NCH16, int fold, mode_bn_epilogue bn,
                                 const float* __restrict__ amax_w) {
  const float sw = F16 ? sp_f16_scale_of(mode::absmax_load(amax_w)) : 1.f;
  const long long total = (long long)d.G * d.MG * NCH16 * KT * MTW * 64;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    long long r = idx >> 6;
    const int m = (int)(r % MTW);
    r /= MTW;
    const int tap = (int)(r % KT);
    r /= KT;
    const int ch = (int)(r % NCH16);
    r /= NCH16;
    const int mg = (int)(r % d.MG);
    const int g = (int)(r / d.MG);
    const int co = mg * 128 + m * 32 + (lane & 31);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = ch * SP_CCH + 8 * (lane >> 5) + j;
      v[j] = 0.f;
      if (co < d.Cog && c < d.Cig) {
        v[j] = w[((long long)(g * d.Cog + co) * d.Cig + c) * KT + tap];
        if (fold) v[j] *= fold_scale(bn, g * d.Cog + co);
        if (F16) v[j] *= sw;
      }
    }
    uint32_t q1[4], q2[4], q3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (F16) {
        sp_split2_f16(v[2 * j], v[2 * j + 1], q1[j], q2[j]);
        q3[j] = 0u;
      } else {
        sp_split2(v[2 * j], v[2 * j + 1], q1[j], q2[j], q3[j]);
      }
    }
    uint4* dst = wps + (idx - lane) * 3 + lane;
    dst[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    dst[64] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
    dst[128] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The TALL-window tiles (next to the poles: windows of 145 rows, or the whole axis with wrap-around) on the same arithmetic (round 4).
// Until then they ran fwd_tile -- fp32 MFMA, matrix-bound, 0.2 ms per tile whatever the image count: the floor of every spherical layer
// of an eval forward at one pair (15 x 0.23 of 11.4 ms) and a third of the training launch.  Their windows do not leave room for an
// operand buffer, so the roles are not split as in the small-tile code below: as in fwd_tile a wave samples its own 32 pixels and
// multiplies them with all four output-channel tiles, straight from registers.  K of one MFMA = 8 input channels x 2 taps (lanes 0..31
// tap 2p, lanes 32..63 tap 2p + 1; five pairs, the second half of the last one multiplies zeros), so a chunk stays 8 channels deep and
// the window staging of fwd_tile is kept as it is.  Per pair and wave: 8 samples per lane (4 LDS words + 4 FMAs each), the exact
// 3-way split, 24 MFMAs against 12 KB of weight fragments (L2 -> L1: what bounds it now: 96 KB per pair and workgroup).
constexpr int TP = 5;  // tap pairs of a 3 x 3 kernel

// wpt[(((((g*MG + mg)*NCH + ch)*TP + pair)*MTW + m)*3 + piece)*64 + lane] = 8 bf16: piece of W[g*Cog + mg*128 + m*32 + (lane&31)]
// [ch*8 + j][tap = 2 pair + (lane>>5)], j = 0..7 (zeros for tap 9; scaled by the folded BatchNorm scale when fold != 0)
template <bool F16>
__global__ void pack_w_win_split_tall(const float* __restrict__ w, uint4* __restrict__ wpt, WinDims d, int fold, mode_bn_epilogue bn,
                                      const float* __restrict__ amax_w) {
  const float sw = F16 ? sp_f16_scale_of(mode::absmax_load(amax_w)) : 1.f;
  const long long total = (long long)d.G * d.MG * d.NCH * TP * MTW * 64;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    long long r = idx >> 6;
    const int m = (int)(r % MTW);
    r /= MTW;
    const int pair = (int)(r % TP);
    r /= TP;
    const int ch = (int)(r % d.NCH);
    r /= d.NCH;
    const int mg = (int)(r % d.MG);
    const int g = (int)(r / d.MG);
    const int co = mg * 128 + m * 32 + (lane & 31);
    const int tap = 2 * pair + (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = ch * CCH + j;
      v[j] = 0.f;
      if (tap < KT && co < d.Cog && c < d.Cig) {
        v[j] = w[((long long)(g * d.Cog + co) * d.Cig + c) * KT + tap];
        if (fold) v[j] *= fold_scale(bn, g * d.Cog + co);
        if (F16) v[j] *= sw;
      }
    }
    uint32_t q1[4], q2[4], q3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (F16) {
        sp_split2_f16(v[2 * j], v[2 * j + 1], q1[j], q2[j]);
        q3[j] = 0u;
      } else {
        sp_split2(v[2 * j], v[2 * j + 1], q1[j], q2[j], q3[j]);
      }
    }
    uint4* dst = wpt + (idx - lane) * 3 + lane;
    dst[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    dst[64] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
    dst[128] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
  }
}

#ifdef MODE_TAPTIME
// debug build only (tools/experiments/sphere_taptime.py): s_memtime stamps of wave 0 of every small-window workgroup
__device__ unsigned long long g_taptime[8 * 8192];
#define MODE_STAMP(j) if (threadIdx.x == 0) g_taptime[((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 8 + (j)] = __builtin_readcyclecounter();
#define MODE_STAMPV(j, v) if (threadIdx.x == 0) g_taptime[((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 8 + (j)] = (v);
#else
#define MODE_STAMP(j)
#define MODE_STAMPV(j, v)
#endif
// Template parameters as fwd_tile (window rows, double-buffered staging in NPH phases of NRB row passes).
// F16: two fp16 pieces / three MFMAs per product (the small-window code of sphere_fwd_split_kernel<false, true>): a half step is 6 slots.
template <int WR_T, bool PIPE, int NRB, int NPH, bool EPI, bool F16 = false>
__device__ __forceinline__ void fwd_tile_split(const float* __restrict__ x, const float* __restrict__ pos, const uint4* __restrict__ wpt,
                                               float* __restrict__ y, const WinDims& d, int h0, int w0, int rbase, int cbase, float* smem,
                                               const Epi& epi, float sx = 1.f, float unscale = 1.f) {
  const int WRP = WR_T > 0 ? WR_T : d.wr;
  const int CP = chan_pitch(WRP);
  const int bufsz = CCH * CP;
  const int b = blockIdx.y;
  const int g = blockIdx.z / d.MG, mg = blockIdx.z % d.MG;
  MODE_STAMP(0)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5;
  const int h = h0 + (wave / TW) * 32 + (lane & 31), w = w0 + (wave % TW);
  const bool pix_ok = h < d.H && w < d.W;
  const long long HW = (long long)d.H * d.W;

  // sampling records of the taps THIS lane samples: tap 2p + half of pair p (tap 9 does not exist: zero weights)
  int roff[TP];
  float4 rw[TP];
  {
    float ph_[TP], pw_[TP];
    const long long idx = pix_ok ? (long long)h * d.W + w : 0;
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const int k = min(2 * p + half, KT - 1);
      ph_[p] = pos[(2 * k) * HW + idx];
      pw_[p] = pos[(2 * k + 1) * HW + idx];
    }
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      int r0 = 0, c0 = 0;
      float4 wt = make_float4(0.f, 0.f, 0.f, 0.f);
      mode::tap_record_fixed(ph_[p], pw_[p], d.H, d.W, r0, c0, wt);
      if (!pix_ok || 2 * p + half >= KT) wt = make_float4(0.f, 0.f, 0.f, 0.f);
      int lr = r0 - rbase;
      if (lr < 0) lr += d.H;
      const int lc = c0 - cbase;
      const bool dead = wt.x == 0.f && wt.y == 0.f && wt.z == 0.f && wt.w == 0.f;
      roff[p] = dead ? 0 : lc * WRP + lr;
      rw[p] = F16 ? make_float4(wt.x * sx, wt.y * sx, wt.z * sx, wt.w * sx) : wt;  // (the operand's power-of-two scale rides on the weights)
    }
  }
  const int lds_floats = (PIPE ? 2 : 1) * bufsz + WRP + 8;
  for (int i = tid; i < lds_floats; i += NTHREADS) smem[i] = 0.f;  // every window word that may be read is finite
  __syncthreads();

  // window staging: exactly fwd_tile's
  const int scol = d.sh == 1 ? tid / SROWS : tid & (WC - 1), srow = d.sh == 1 ? tid % SROWS : tid / WC;
  const int gcol = cbase + scol;
  const bool col_ok = gcol < d.W;
  const float* xg = x + ((long long)b * d.Ci + (long long)g * d.Cig) * HW + (col_ok ? gcol * d.sw : 0);
  int rowoff[NRB];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) {
    const int r = rb * SROWS + srow;
    rowoff[rb] = ((rbase + (r < WRP ? r : 0)) % d.H) * d.sh;
  }
  constexpr int CPH = CCH / NPH;
  float lv[PIPE ? CPH * NRB : 1];
  auto issue = [&](int ch, int ph) {
#pragma unroll
    for (int cc = 0; cc < CPH; ++cc) {
      const int chan = ch * CCH + ph * CPH + cc;
      const float* xc = xg + (long long)(chan < d.Cig ? chan : 0) * HW;
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) lv[cc * NRB + rb] = xc[rowoff[rb]];
    }
  };
  auto commit = [&](int ch, int ph, float* buf) {
#pragma unroll
    for (int cc = 0; cc < CPH; ++cc) {
      const int c = ph * CPH + cc;
      const bool ok = col_ok && (ch * CCH + c < d.Cig);
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        const int r = rb * SROWS + srow;
        if (r < WRP) buf[c * CP + scol * WRP + r] = ok ? lv[cc * NRB + rb] : 0.f;
      }
    }
  };
  f32x16 acc[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  // WEIGHT FRAGMENTS THROUGH LDS (round 4).  A pair step multiplies this wave's B fragment with all four output tiles: 12 fragments =
  // 12 KB, the same for the 8 waves.  Fetched by every wave on its own they were 96 KB per step and workgroup through the vector
  // memory path (L1 holds 32 KB; a chunk's five steps are 60 KB), and a tall tile took 0.18 ms against 0.09 ms for a small-window
  // tile -- the critical path of the launch at 2 and 4 images.  Now the workgroup fetches a step's 12 KB once (1.5 x 16 B per thread),
  // one step ahead, into one of two LDS buffers behind the windows, and the waves read their fragments from there.  One barrier per
  // step, in its middle: first half = tiles 0, 1 (fragments read during the second half of the previous step), second half = tiles
  // 2, 3 (read during the first half); the next step's weights are stored at the top of a step and visible after its barrier.
  // The sampling arithmetic of the next step sits between the MFMAs slot by slot (see sphere_fwd_split_kernel).
  static_assert(PIPE, "the split tall tiles are double-buffered");
  constexpr int WSTEP = MTW * 3 * 64;  // uint4 per pair step
  uint4* wbuf = reinterpret_cast<uint4*>(smem + ((lds_floats + 3) / 4) * 4);  // [2][WSTEP]
  const uint4* wsrc = wpt + ((long long)(g * d.MG + mg) * d.NCH) * TP * WSTEP;  // + step * WSTEP + fragment * 64 + lane
  const int nsteps = d.NCH * TP;
  uint4 wg0, wg1;  // the 1.5 x 16 bytes this thread fetches of the step after the next
  auto wfetch = [&](int step) {
    const uint4* src = wsrc + (long long)(step < nsteps ? step : nsteps - 1) * WSTEP;
    wg0 = src[tid];
    wg1 = src[NTHREADS + (tid & 255)];
  };
  auto wstore = [&](int step) {
    uint4* dst = wbuf + (step & 1) * WSTEP;
    dst[tid] = wg0;
    if (tid < 256) dst[NTHREADS + tid] = wg1;
  };
  wfetch(0);

  // the 8 channel values of this lane's (pixel, tap) for pair p: 32 window words, then 4 FMAs each and the split
  float raw[16], v[8], ra[4], rb[4];  // window words of HALF a sample set (4 channels), requested in two batches per step
  uint32_t q1[4], q2[4], q3[4];
  auto load_half = [&](const float* buf, int p, int hb) {
    const float* q0 = buf + roff[p] + hb * 4 * CP;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float* q = q0 + c * CP;
      raw[c * 4 + 0] = q[0];
      raw[c * 4 + 1] = q[WRP];
      raw[c * 4 + 2] = q[1];
      raw[c * 4 + 3] = q[WRP + 1];
    }
  };
  auto comb = [&](int p, int c) {  // (one scalar chain per value: see sphere_fwd_split_kernel)
    const float4 tw = rw[p];
    const int r = (c & 3) * 4;
    v[c] = __builtin_fmaf(tw.w, raw[r + 3], __builtin_fmaf(tw.z, raw[r + 2], __builtin_fmaf(tw.y, raw[r + 1], tw.x * raw[r])));
    asm("" : "+v"(v[c]));
  };
  auto split_a = [&](int j) {
    q1[j] = sp_pack2(v[2 * j], v[2 * j + 1]);
    ra[j] = v[2 * j] - __builtin_bit_cast(float, q1[j] << 16);
    rb[j] = v[2 * j + 1] - __builtin_bit_cast(float, q1[j] & 0xffff0000u);
    asm("" : "+v"(ra[j]), "+v"(rb[j]));
  };
  auto split_b = [&](int j) {
    q2[j] = sp_pack2(ra[j], rb[j]);
    ra[j] = ra[j] - __builtin_bit_cast(float, q2[j] << 16);
    rb[j] = rb[j] - __builtin_bit_cast(float, q2[j] & 0xffff0000u);
    asm("" : "+v"(ra[j]), "+v"(rb[j]));
  };
  auto split_c = [&](int j) { q3[j] = sp_pack2(ra[j], rb[j]); };
  auto hsplit = [&](int j) { sp_split2_f16(v[2 * j], v[2 * j + 1], q1[j], q2[j]); };

#pragma unroll
  for (int ph = 0; ph < NPH; ++ph) {
    issue(0, ph);
    commit(0, ph, smem);
  }
  wstore(0);
  wfetch(1);
  __syncthreads();
  uint4 a_cur[2][3], a_nxt[2][3], bq[3];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 3; ++q) a_cur[m][q] = wbuf[(m * 3 + q) * 64 + lane];
  constexpr int PPP = TP / NPH;  // tap pairs per staging phase (NPH = 2: pairs 0-1 | 2-4; NPH = 4: one pair each, the last phase two)
  // B fragment of the first pair of the first chunk; every later one is built under the MFMAs of the step before it -- the first pair
  // of a chunk too: the last phase of the next chunk's window is stored in the FIRST half of the chunk's last step, so that window is
  // complete at that step's barrier and its second half can sample from it (no chunk-top sampling, no barrier at the chunk end)
  load_half(smem, 0, 0);
#pragma unroll
  for (int c = 0; c < 4; ++c) comb(0, c);
  load_half(smem, 0, 1);
#pragma unroll
  for (int c = 4; c < 8; ++c) comb(0, c);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if constexpr (F16) {
      hsplit(j);
      q3[j] = 0u;
    } else {
      split_a(j);
      split_b(j);
      split_c(j);
    }
  }
  bq[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
  bq[1] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
  bq[2] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
  load_half(smem, 1, 0);  // first half batch of the fragment built under step 0
  MODE_STAMP(1)
#define MODE_SB __builtin_amdgcn_sched_barrier(0);
#define MODE_TMF(PA, PB, m, t) acc[t] = sp_mfma(a_cur[m][PA], bq[PB], acc[t]);
#define MODE_TMFH(PA, PB, m, t) acc[t] = sp_mfma_f16(a_cur[m][PA], bq[PB], acc[t]); asm volatile("" :: "v"(acc[t]));  // (pinned: see the small tiles)
#ifdef MODE_TAPTIME
  unsigned long long tt_top = 0, tt_pre = 0, tt_post = __builtin_readcyclecounter(), ts_a = 0, ts_b = 0, ts_c = 0;
#endif
  for (int ch = 0; ch < d.NCH; ++ch) {
    float* cur = smem + (ch & 1) * bufsz;
    float* nxt = smem + ((ch + 1) & 1) * bufsz;
    const bool more = ch + 1 < d.NCH;
#pragma unroll
    for (int p = 0; p < TP; ++p) {
      const int step = ch * TP + p;
      const uint4* wcur = wbuf + (step & 1) * WSTEP + lane;
      const uint4* wnx = wbuf + ((step + 1) & 1) * WSTEP + lane;
      const int pn = p + 1 < TP ? p + 1 : 0;  // the pair whose B fragment is built under this step (pair 0 of the next chunk under the last)
      // ---- first half: output tiles 0, 1.  LDS traffic is spread over the slots (requested in one batch at the top, the 24 reads and
      // stores of the 8 waves backed up in front of the first MFMA: 1 650 cycles for this half against 930 for the other)
#ifdef MODE_TAPTIME
      tt_top = __builtin_readcyclecounter();
#endif
      if constexpr (F16) {
        // 6 slots: lo x hi, hi x lo, hi x hi for tiles 0 and 1; the rest of the step's LDS and sampling work between them as below
        MODE_SB
        MODE_TMFH(1, 0, 0, 0) if (p + 1 < TP) comb(pn, 0); MODE_SB
        MODE_TMFH(1, 0, 1, 1) if (p + 1 < TP) comb(pn, 1); MODE_SB
        MODE_TMFH(0, 1, 0, 0) if (p + 1 < TP) comb(pn, 2); wstore(step + 1); MODE_SB
        MODE_TMFH(0, 1, 1, 1) if (p + 1 < TP) comb(pn, 3); wfetch(step + 2); MODE_SB
        if (p + 1 < TP) load_half(cur, pn, 1);
        if (p % PPP == 0 && p / PPP < NPH && more) issue(ch + 1, p / PPP);
        if (p == TP - 1 && more) commit(ch + 1, NPH - 1, nxt);
#pragma unroll
        for (int q = 0; q < 2; ++q) a_nxt[0][q] = wcur[((2 + 0) * 3 + q) * 64];
#pragma unroll
        for (int q = 0; q < 2; ++q) a_nxt[1][q] = wcur[((2 + 1) * 3 + q) * 64];
        MODE_SB
        MODE_TMFH(0, 0, 0, 0) if (p + 1 < TP) { comb(pn, 4); comb(pn, 5); } MODE_SB
        MODE_TMFH(0, 0, 1, 1) if (p + 1 < TP) { comb(pn, 6); comb(pn, 7); } MODE_SB
      } else {
      MODE_SB
      MODE_TMF(2, 0, 0, 0) if (p + 1 < TP) comb(pn, 0); MODE_SB
      MODE_TMF(2, 0, 1, 1) if (p + 1 < TP) comb(pn, 1); MODE_SB
      MODE_TMF(0, 2, 0, 0) if (p + 1 < TP) comb(pn, 2); wstore(step + 1); MODE_SB
      MODE_TMF(0, 2, 1, 1) if (p + 1 < TP) comb(pn, 3); wfetch(step + 2); MODE_SB
      if (p + 1 < TP) load_half(cur, pn, 1);
      if (p % PPP == 0 && p / PPP < NPH && more) issue(ch + 1, p / PPP);  // rows of the next chunk fly under the MFMAs below
      if (p == TP - 1 && more) commit(ch + 1, NPH - 1, nxt);  // (its loads were requested a step ago)
      MODE_SB
      MODE_TMF(1, 1, 0, 0) MODE_SB
#pragma unroll
      for (int q = 0; q < 3; ++q) a_nxt[0][q] = wcur[((2 + 0) * 3 + q) * 64];
      MODE_SB
      MODE_TMF(1, 1, 1, 1) MODE_SB
#pragma unroll
      for (int q = 0; q < 3; ++q) a_nxt[1][q] = wcur[((2 + 1) * 3 + q) * 64];
      MODE_SB
      MODE_TMF(1, 0, 0, 0) if (p + 1 < TP) comb(pn, 4); MODE_SB
      MODE_TMF(1, 0, 1, 1) if (p + 1 < TP) comb(pn, 5); MODE_SB
      MODE_TMF(0, 1, 0, 0) if (p + 1 < TP) comb(pn, 6); MODE_SB
      MODE_TMF(0, 1, 1, 1) if (p + 1 < TP) comb(pn, 7); MODE_SB
      MODE_TMF(0, 0, 0, 0) MODE_SB
      MODE_TMF(0, 0, 1, 1) MODE_SB
      }
#ifdef MODE_TAPTIME
      tt_pre = __builtin_readcyclecounter();
#endif
      sp_lds_barrier();  // the weights of step + 1 are in LDS
#ifdef MODE_TAPTIME
      ts_c += tt_top - tt_post;  // second half of the previous step
      tt_post = __builtin_readcyclecounter();
      ts_a += tt_pre - tt_top;
      ts_b += tt_post - tt_pre;
#endif
      // ---- second half: output tiles 2, 3
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 3; ++q) a_cur[m][q] = a_nxt[m][q];
      if (p + 1 == TP) load_half(more ? nxt : cur, 0, 0);  // (after the last chunk: any finite words, the fragment is not used)
      MODE_SB
      if constexpr (F16) {
        if (p + 1 < TP) {
          MODE_TMFH(1, 0, 0, 2) hsplit(0); MODE_SB
          MODE_TMFH(1, 0, 1, 3) hsplit(1); MODE_SB
#pragma unroll
          for (int q = 0; q < 2; ++q) a_nxt[0][q] = wnx[(0 * 3 + q) * 64];
          MODE_SB
          MODE_TMFH(0, 1, 0, 2) hsplit(2); MODE_SB
          MODE_TMFH(0, 1, 1, 3) hsplit(3); MODE_SB
#pragma unroll
          for (int q = 0; q < 2; ++q) a_nxt[1][q] = wnx[(1 * 3 + q) * 64];
          if (p + 2 < TP) load_half(cur, p + 2, 0);  // first half batch of the fragment built under the next step
          MODE_SB
          MODE_TMFH(0, 0, 0, 2) MODE_SB
          MODE_TMFH(0, 0, 1, 3) MODE_SB
        } else {  // the whole B fragment of the next chunk's first pair under these six MFMAs
          MODE_TMFH(1, 0, 0, 2) comb(0, 0); comb(0, 1); MODE_SB
          MODE_TMFH(1, 0, 1, 3) comb(0, 2); comb(0, 3); MODE_SB
          load_half(more ? nxt : cur, 0, 1);
#pragma unroll
          for (int q = 0; q < 2; ++q) a_nxt[0][q] = wnx[(0 * 3 + q) * 64];
#pragma unroll
          for (int q = 0; q < 2; ++q) a_nxt[1][q] = wnx[(1 * 3 + q) * 64];
          MODE_SB
          MODE_TMFH(0, 1, 0, 2) MODE_SB
          MODE_TMFH(0, 1, 1, 3) comb(0, 4); comb(0, 5); comb(0, 6); comb(0, 7); MODE_SB
          load_half(more ? nxt : cur, 1, 0);  // (the fragment built under the next chunk's first step)
          MODE_SB
          MODE_TMFH(0, 0, 0, 2) hsplit(0); hsplit(1); MODE_SB
          MODE_TMFH(0, 0, 1, 3) hsplit(2); hsplit(3); MODE_SB
        }
      } else
      if (p + 1 < TP) {
        MODE_TMF(2, 0, 0, 2) split_a(0); MODE_SB
        MODE_TMF(2, 0, 1, 3) split_b(0); MODE_SB
#pragma unroll
        for (int q = 0; q < 3; ++q) a_nxt[0][q] = wnx[(0 * 3 + q) * 64];
        MODE_SB
        MODE_TMF(0, 2, 0, 2) split_c(0); split_a(1); MODE_SB
        MODE_TMF(0, 2, 1, 3) split_b(1); MODE_SB
#pragma unroll
        for (int q = 0; q < 3; ++q) a_nxt[1][q] = wnx[(1 * 3 + q) * 64];
        MODE_SB
        MODE_TMF(1, 1, 0, 2) split_c(1); split_a(2); MODE_SB
        MODE_TMF(1, 1, 1, 3) split_b(2); MODE_SB
        MODE_TMF(1, 0, 0, 2) split_c(2); split_a(3); MODE_SB
        MODE_TMF(1, 0, 1, 3) split_b(3); MODE_SB
        MODE_TMF(0, 1, 0, 2) split_c(3); MODE_SB
        if (p + 2 < TP) load_half(cur, p + 2, 0);  // first half batch of the fragment built under the next step
        MODE_SB
        MODE_TMF(0, 1, 1, 3) MODE_SB
        MODE_TMF(0, 0, 0, 2) MODE_SB
        MODE_TMF(0, 0, 1, 3) MODE_SB
      } else {  // the whole B fragment of the next chunk's first pair under these twelve MFMAs
        MODE_TMF(2, 0, 0, 2) MODE_SB
        MODE_TMF(2, 0, 1, 3) MODE_SB
        MODE_TMF(0, 2, 0, 2) comb(0, 0); comb(0, 1); MODE_SB
        MODE_TMF(0, 2, 1, 3) comb(0, 2); comb(0, 3); MODE_SB
        load_half(more ? nxt : cur, 0, 1);
#pragma unroll
        for (int q = 0; q < 3; ++q) a_nxt[0][q] = wnx[(0 * 3 + q) * 64];
        MODE_SB
        MODE_TMF(1, 1, 0, 2) MODE_SB
#pragma unroll
        for (int q = 0; q < 3; ++q) a_nxt[1][q] = wnx[(1 * 3 + q) * 64];
        MODE_SB
        MODE_TMF(1, 1, 1, 3) MODE_SB
        MODE_TMF(1, 0, 0, 2) comb(0, 4); comb(0, 5); MODE_SB
        MODE_TMF(1, 0, 1, 3) comb(0, 6); comb(0, 7); MODE_SB
        load_half(more ? nxt : cur, 1, 0);  // (the fragment built under the next chunk's first step)
        MODE_SB
        MODE_TMF(0, 1, 0, 2) split_a(0); split_a(1); split_a(2); MODE_SB
        MODE_TMF(0, 1, 1, 3) split_a(3); split_b(0); split_b(1); MODE_SB
        MODE_TMF(0, 0, 0, 2) split_b(2); split_b(3); split_c(0); MODE_SB
        MODE_TMF(0, 0, 1, 3) split_c(1); split_c(2); split_c(3); MODE_SB
      }
      if (more && p < TP - 1 && p % PPP == PPP - 1 && p / PPP < NPH - 1) commit(ch + 1, p / PPP, nxt);  // the other buffer: nobody reads it now
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 3; ++q) a_cur[m][q] = a_nxt[m][q];
      bq[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
      bq[1] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
      bq[2] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
    }
  }
#undef MODE_SB
#undef MODE_TMF
#undef MODE_TMFH
  MODE_STAMP(2)
#ifdef MODE_TAPTIME
  MODE_STAMPV(4, ts_a)
  MODE_STAMPV(5, ts_b)
  MODE_STAMPV(6, ts_c)
#endif

  if (pix_ok) {
    float* yb = y + ((long long)b * d.Co + (long long)g * d.Cog + (long long)mg * 128) * HW + (long long)h * d.sh + (long long)w * d.sw;
    const int cmax = d.Cog - mg * 128;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      // (eval mode: the 16 shifts and the 16 residual values of this M-tile are requested TOGETHER ahead of its stores, under one uniform
      // test each: read inside the store loop every store waited for a load of its own -- see sphere_fwd_split_kernel's epilogue)
      float res[16], shv[16];
      if (EPI) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = min(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, cmax - 1);
          shv[r] = epi.shift[g * d.Cog + mg * 128 + co];
          res[r] = 0.f;
        }
        if (epi.add) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = min(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, cmax - 1);
            res[r] = epi.add[(yb - y) + (long long)co * HW];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      auto emit = [&](int r) {
        const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (EPI) {
          const float v = (acc[m][r] + shv[r]) + res[r];
          yb[(long long)co * HW] = epi.relu ? relu_nan(v) : v;
        } else {
          yb[(long long)co * HW] = F16 ? acc[m][r] * unscale : acc[m][r];
        }
      };
      if (m * 32 + 32 <= cmax) {  // (uniform) a full M-tile: no test per store
#pragma unroll
        for (int r = 0; r < 16; ++r) emit(r);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < cmax) emit(r);
      }
      if (EPI) __builtin_amdgcn_sched_barrier(0);
    }
  }
  MODE_STAMP(3)
}

// F16 (training forward, no epilogue): two fp16 pieces and three MFMAs per product -- a tap of the small-window tiles is 12 slots
// instead of 24, and the split of a sampled value 3 instructions instead of 5.5; the tall-window tiles (fwd_tile_split<.., F16>) likewise.
template <bool EPI, bool F16 = false>
__global__ __launch_bounds__(NTHREADS) void sphere_fwd_split_kernel(const float* __restrict__ x, const float* __restrict__ pos,
                                                                    const uint4* __restrict__ wps, const uint4* __restrict__ wpt,
                                                                    float* __restrict__ y, WinDims d, int NCH16,
                                                                    const int4* __restrict__ tiles, Epi epi,
                                                                    const float* __restrict__ amax_x, const float* __restrict__ amax_w) {
  static_assert(!(EPI && F16), "the fp16 arithmetic has no folded-BatchNorm epilogue");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WRP = WR_SMALL, CP = SP_CP;
  uint4* opbuf = reinterpret_cast<uint4*>(smem + SP_WIN_FLOATS);  // [3][8 pixel groups][3 pieces][64 lanes]
  const int4 t = tiles[blockIdx.x];
  const int h0 = t.x, w0 = t.y, rbase = t.z, cbase = t.w & 0xffff;
  // the few tall-window tiles (next to the poles) run INSIDE this launch (as a launch of their own they would be an under-filled tail:
  // 0.37 ms for the two launches against 0.26 for the old single one), on the same arithmetic since round 4 (fwd_tile_split)
  const int cls = t.w >> 16;
  if (cls == 1) {
    if constexpr (F16) {
      const float sx_ = sp_f16_scale_of(mode::absmax_load(amax_x));
      fwd_tile_split<WR_MID, true, (WR_MID + SROWS - 1) / SROWS, 2, EPI, true>(x, pos, wpt, y, d, t.x, t.y, t.z, cbase, smem, epi, sx_,
                                                                              (1.f / sx_) * (1.f / sp_f16_scale_of(mode::absmax_load(amax_w))));
    } else {
      fwd_tile_split<WR_MID, true, (WR_MID + SROWS - 1) / SROWS, 2, EPI>(x, pos, wpt, y, d, t.x, t.y, t.z, cbase, smem, epi);
    }
    return;
  }
  if (cls != 0) {
    if constexpr (F16) {
      const float sx_ = sp_f16_scale_of(mode::absmax_load(amax_x));
      fwd_tile_split<0, true, WR_PIPE_MAX / SROWS, 4, EPI, true>(x, pos, wpt, y, d, t.x, t.y, t.z, cbase, smem, epi, sx_,
                                                                 (1.f / sx_) * (1.f / sp_f16_scale_of(mode::absmax_load(amax_w))));
    } else {
      fwd_tile_split<0, true, WR_PIPE_MAX / SROWS, 4, EPI>(x, pos, wpt, y, d, t.x, t.y, t.z, cbase, smem, epi);
    }
    return;
  }
  MODE_STAMP(0)
  const int b = blockIdx.y;
  const int g = blockIdx.z / d.MG, mg = blockIdx.z % d.MG;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5;
  const int h = h0 + (wave / TW) * 32 + (lane & 31), w = w0 + (wave % TW);  // the pixel this lane SAMPLES
  const bool pix_ok = h < d.H && w < d.W;
  const long long HW = (long long)d.H * d.W;
  float sx = 1.f, unscale = 1.f;
  if (F16) {
    sx = sp_f16_scale_of(mode::absmax_load(amax_x));
    unscale = (1.f / sx) * (1.f / sp_f16_scale_of(mode::absmax_load(amax_w)));
  }

  int roff[KT];
  float4 rw[KT];
  // (all 18 table values first, from a clamped index: inside `if (pix_ok)` they were nine dependent load -> compute rounds, and the
  // prologue of a tile is not overlapped with anything)
  float ph_[KT], pw_[KT];
  {
    const long long idx = pix_ok ? (long long)h * d.W + w : 0;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      ph_[k] = pos[(2 * k) * HW + idx];
      pw_[k] = pos[(2 * k + 1) * HW + idx];
    }
  }
  // every window word that may be read is finite: the staging stores write all rows and columns of both windows; what they never
  // write is the pad word behind each channel and the slack behind the two buffers (read only under zero weights)
  if (tid < 2 * SP_CCH) smem[(tid / SP_CCH) * SP_WIN + (tid % SP_CCH) * CP + WC * WRP] = 0.f;
  if (tid < WRP + 8) smem[SP_WIN + tid] = 0.f;  // (the last channel of the first window runs over into the second, which is empty during the first chunk)
  for (int i = 2 * SP_WIN + tid; i < SP_WIN_FLOATS; i += NTHREADS) smem[i] = 0.f;
  // (no barrier here: it would drain the table loads above before the first window is even requested -- 9 k of a workgroup's 190 k
  // cycles.  The only words both this fill and the staging stores below write are slack words, where either value will do; the barrier
  // behind the staging stores orders everything in front of the first sample)
  MODE_STAMP(4)

  // window staging: thread -> (window column, row within a pass of SROWS rows), as in fwd_tile; 2 passes cover the 81 rows
  const int scol = d.sh == 1 ? tid / SROWS : tid & (WC - 1), srow = d.sh == 1 ? tid % SROWS : tid / WC;
  const int gcol = cbase + scol;
  const bool col_ok = gcol < d.W;
  const float* xg = x + ((long long)b * d.Ci + (long long)g * d.Cig) * HW;  // uniform; lanes add 32-bit element offsets (H * W < 2^30)
  unsigned rowoff[2];  // BYTE offsets inside a channel plane
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int r = rb * SROWS + srow;
    rowoff[rb] = 4u * (unsigned)(((rbase + (r < WRP ? r : 0)) % d.H) * d.sh + (col_ok ? gcol * d.sw : 0));
  }
  const long long plane_bytes = 4 * HW;
  constexpr int STAGE_LAG = 2;  // taps between the loads of a staging phase and its LDS stores
  float lv[8][4];  // the 8 staging phases of a chunk (phase = 2 channels x 2 row passes), all in flight at once
  auto issue = [&](int ch, int ph, int set) {  // (Cig is a multiple of 16: every channel of a chunk exists)
    const char* xc = reinterpret_cast<const char*>(xg) + ((long long)ch * SP_CCH + ph * 2) * plane_bytes;  // uniform
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) lv[set][cc * 2 + rb] = *reinterpret_cast<const float*>(xc + cc * plane_bytes + rowoff[rb]);
  };
  auto commit = [&](int ch, int ph, int set, float* buf) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int c = ph * 2 + cc;
      const bool ok = col_ok;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        // rows beyond the window go to the slack words behind the two buffers (finite values, read only under zero weights): an
        // `if` here is a branch, and a branch in the middle of a tap pins the sampling code behind the MFMAs
        const int r = rb * SROWS + srow;
        float* dst = r < WRP ? buf + c * CP + scol * WRP + r : smem + 2 * SP_WIN + (tid & 7);
        *dst = ok ? lv[set][cc * 2 + rb] : 0.f;
      }
    }
  };
  // B fragment of this lane's pixel for tap k of the chunk in `win`: 8 channels, split, stored as this wave's group.  In two halves:
  // sample_read requests the 32 window words (16 ds_read2_b32) in ONE batch, sample_finish combines, splits and stores them.  As one
  // lambda the compiler issued each read right in front of its use (`ds_read2_b32; s_waitcnt lgkmcnt(0)` sixteen times per tap, all
  // of it BEHIND the tap's 24 MFMAs): sixteen serialised LDS round trips per tap and wave, 2 700 cycles per tap against the 1 536 the
  // matrix pipe needs (tools/experiments/tap_pipeline.hip reproduces it without the rest of the kernel)
  auto sample_read = [&](const float* win, int k, float (&raw)[8][4]) {
    const float* p = win + half * 8 * CP + roff[k];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float* q = p + c * CP;
      raw[c][0] = q[0];
      raw[c][1] = q[WRP];
      raw[c][2] = q[1];
      raw[c][3] = q[WRP + 1];
    }
  };
  auto sample_finish = [&](int k, float (&raw)[8][4], uint4* op) {
    const float4 tw = rw[k];
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      // (one fma chain per value: left to itself the compiler SLP-packs these into v_pk_fma_f32, which costs the MFMA stream more
      // than two plain FMAs)
      v[c] = __builtin_fmaf(tw.w, raw[c][3], __builtin_fmaf(tw.z, raw[c][2], __builtin_fmaf(tw.y, raw[c][1], tw.x * raw[c][0])));
      asm("" : "+v"(v[c]));  // (opaque, not volatile -- a volatile asm pins the schedule: keeps the chains of two channels from being packed pairwise -- 24 v_mov + 24 v_pk_*)
    }
    uint32_t q1[4], q2[4], q3[4];
    uint4* dst = op + (wave * 3) * 64 + lane;
    if constexpr (F16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) sp_split2_f16(v[2 * j], v[2 * j + 1], q1[j], q2[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) sp_split2(v[2 * j], v[2 * j + 1], q1[j], q2[j], q3[j]);
      dst[128] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
    }
    dst[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    dst[64] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
  };
  auto sample = [&](const float* win, int k, uint4* op) {
    float raw[8][4];
    sample_read(win, k, raw);
    __builtin_amdgcn_sched_barrier(0);  // (all 16 reads requested before the first one is used)
    sample_finish(k, raw, op);
  };

  f32x16 acc[4];
#pragma unroll
  for (int gi = 0; gi < 4; ++gi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[gi][r] = 0.f;
  const int m = wave % TW, gset = (wave / TW) * 4;  // output tile and first pixel group of the MATRIX role
  const uint4* wpa = wps + ((long long)(g * d.MG + mg) * NCH16) * KT * MTW * 192 + m * 192;  // uniform; + p * 64 + lane per fragment
  const int nsteps = NCH16 * KT;
  constexpr int NPC = F16 ? 2 : 3;  // pieces per value
  uint4 acur[3], anxt[3];  // weight fragments of this tap and of the next one (requested at the top of a tap, one tap ahead)
#pragma unroll
  for (int p = 0; p < NPC; ++p) acur[p] = wpa[(unsigned)(p * 64 + lane)];

  // prologue: window of chunk 0, operand of tap 0
#pragma unroll
  for (int ph = 0; ph < 8; ++ph) issue(0, ph, ph);  // 32 loads in flight, then the stores: one memory round trip, not eight
  // (the sampling records are computed while those loads fly)
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    int r0 = 0, c0 = 0;
    float4 wt = make_float4(0.f, 0.f, 0.f, 0.f);
    mode::tap_record_fixed(ph_[k], pw_[k], d.H, d.W, r0, c0, wt);
    if (!pix_ok) wt = make_float4(0.f, 0.f, 0.f, 0.f);
    int lr = r0 - rbase;
    if (lr < 0) lr += d.H;
    const int lc = c0 - cbase;
    const bool dead = wt.x == 0.f && wt.y == 0.f && wt.z == 0.f && wt.w == 0.f;
    roff[k] = (dead || !pix_ok) ? 0 : lc * WRP + lr;
    // (F16: the operand's power-of-two scale rides on the four weights -- sx * (sum of w_i x_i) bit for bit, no instruction per sample)
    rw[k] = F16 ? make_float4(wt.x * sx, wt.y * sx, wt.z * sx, wt.w * sx) : wt;
  }
  MODE_STAMP(5)
#pragma unroll
  for (int ph = 0; ph < 8; ++ph) commit(0, ph, ph, smem);
  __syncthreads();
  MODE_STAMP(6)
  // ---- the tap loop.  Three operand buffers: tap t multiplies fragments of buffer t % 3, samples tap t + 2 into buffer (t + 2) % 3 and
  // reads piece 0 of tap t + 1 near its end, so no MFMA waits for an LDS read of its own tap.  A tap is 24 SLOTS (one MFMA each) with
  // everything else placed by hand between them and fenced with sched_barrier(0): left to the scheduler, the window reads ended up
  // one by one in front of their uses, behind all 24 MFMAs (see sample_read above).  Fragment registers: piece 0 (slots 0..11), piece
  // 1 (read in slot 4, used 12..19), piece 2 (read in slot 12, used 20..23), piece 0 of the next tap (read in slot 20) -- never more
  // than two of the four sets alive.  Window words: two half batches of 4 channels (16 registers), the second requested in slot 4
  // when the first has been consumed.  Measured on the skeleton of this loop (tools/experiments/tap_pipeline.hip, gen_tap_asm.py):
  // 2 750 cycles per tap as it was, ~2 100 in this form, 1 500 for the MFMAs alone.
  sample(smem, 0, opbuf);
  sample(smem, 1, opbuf + SP_OP);
  sp_lds_barrier();
  uint4 b0[4], b0n[4], b1[4], b2[4];
  const uint4* fragbase = opbuf + gset * 192 + lane;  // + buffer * SP_OP + (gi * 3 + piece) * 64
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) b0[gi] = fragbase[(gi * 3 + 0) * 64];
  float raw[4][4], v[8], ra[4], rb[4];
  uint32_t q1[4], q2[4], q3[4];
  {
    const float* wp0 = smem + half * 8 * CP + roff[2];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float* q = wp0 + c * CP;
      raw[c][0] = q[0];
      raw[c][1] = q[WRP];
      raw[c][2] = q[1];
      raw[c][3] = q[WRP + 1];
    }
  }
  MODE_STAMP(1)
#define MODE_SB __builtin_amdgcn_sched_barrier(0);
  if constexpr (F16) {
    // ---- 12 slots: lo x hi (piece 1 of the weights against piece 0 of the operand), hi x hi, hi x lo.  The second half batch of window
    // words has registers of its own (the third piece's are free) and is requested at the top of the tap; the next tap's first half batch
    // and its piece-0 fragments in slot 8, when the combines have consumed this tap's.
#define MODE_MFH(PA, B, gi) acc[gi] = sp_mfma_f16(acur[PA], B[gi], acc[gi]); asm volatile("" :: "v"(acc[gi]));
    float raw2[4][4];
    for (int ch = 0; ch < NCH16; ++ch) {
      float* cur = smem + (ch & 1) * SP_WIN;
      float* nxt = smem + ((ch + 1) & 1) * SP_WIN;
      const int chn = ch + 1 < NCH16 ? ch + 1 : ch;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const int step = ch * KT + k;
        const int nstep = step + 1 < nsteps ? step + 1 : nsteps - 1;
        const int ks = (k + 2) % KT;
        const float* wsrc = k + 2 < KT ? cur : nxt;
        const uint4* fcur = fragbase + (k % 3) * SP_OP;
        const uint4* fnxt = fragbase + ((k + 1) % 3) * SP_OP;
        uint4* opw = opbuf + ((k + 2) % 3) * SP_OP + (wave * 3) * 64 + lane;
        const float4 tw = rw[ks];
        const float* wp_ = wsrc + half * 8 * CP + roff[ks];
        const float* wpn_ = (k + 3 < KT ? cur : nxt) + half * 8 * CP + roff[(k + 3) % KT];
        auto combine = [&](int c, float (&rr)[4][4]) {  // (one fma chain per value, kept scalar: see sp_split2)
          v[c] = __builtin_fmaf(tw.w, rr[c & 3][3], __builtin_fmaf(tw.z, rr[c & 3][2], __builtin_fmaf(tw.y, rr[c & 3][1], tw.x * rr[c & 3][0])));
          asm("" : "+v"(v[c]));
        };
        auto hsplit = [&](int j) { sp_split2_f16(v[2 * j], v[2 * j + 1], q1[j], q2[j]); };
#pragma unroll
        for (int p = 0; p < 2; ++p) anxt[p] = (wpa + (long long)nstep * MTW * 192)[(unsigned)(p * 64 + lane)];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float* q = wp_ + (4 + c) * CP;
          raw2[c][0] = q[0];
          raw2[c][1] = q[WRP];
          raw2[c][2] = q[1];
          raw2[c][3] = q[WRP + 1];
        }
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) b1[gi] = fcur[(gi * 3 + 1) * 64];
        MODE_SB
        MODE_MFH(0, b0, 0) combine(0, raw); MODE_SB
        MODE_MFH(0, b0, 1) combine(1, raw); MODE_SB
        MODE_MFH(0, b0, 2) combine(2, raw); MODE_SB
        MODE_MFH(0, b0, 3) combine(3, raw); MODE_SB
        MODE_MFH(1, b0, 0) hsplit(0); MODE_SB
        MODE_MFH(1, b0, 1) hsplit(1); combine(4, raw2); MODE_SB
        MODE_MFH(1, b0, 2) combine(5, raw2); combine(6, raw2); MODE_SB
        MODE_MFH(1, b0, 3) combine(7, raw2); MODE_SB
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) b0n[gi] = fnxt[(gi * 3 + 0) * 64];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float* q = wpn_ + c * CP;
          raw[c][0] = q[0];
          raw[c][1] = q[WRP];
          raw[c][2] = q[1];
          raw[c][3] = q[WRP + 1];
        }
        MODE_SB
        MODE_MFH(0, b1, 0) hsplit(2); MODE_SB
        MODE_MFH(0, b1, 1) hsplit(3); MODE_SB
        opw[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
        opw[64] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
        if (k < 4) {
          issue(chn, 2 * k, 2 * k);
          issue(chn, 2 * k + 1, 2 * k + 1);
        }
        MODE_SB
        MODE_MFH(0, b1, 2) MODE_SB
        if (k >= STAGE_LAG && k < 4 + STAGE_LAG) commit(chn, 2 * (k - STAGE_LAG), 2 * (k - STAGE_LAG), nxt);
        MODE_SB
        MODE_MFH(0, b1, 3) MODE_SB
        if (k >= STAGE_LAG && k < 4 + STAGE_LAG) commit(chn, 2 * (k - STAGE_LAG) + 1, 2 * (k - STAGE_LAG) + 1, nxt);
        MODE_SB
#pragma unroll
        for (int p = 0; p < 2; ++p) acur[p] = anxt[p];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) b0[gi] = b0n[gi];
        sp_lds_barrier();
      }
    }
#undef MODE_MFH
  } else {
#define MODE_MF(PA, B, gi) acc[gi] = sp_mfma(acur[PA], B[gi], acc[gi]);

  for (int ch = 0; ch < NCH16; ++ch) {
    float* cur = smem + (ch & 1) * SP_WIN;
    float* nxt = smem + ((ch + 1) & 1) * SP_WIN;
    const int chn = ch + 1 < NCH16 ? ch + 1 : ch;  // (after the last chunk the same window is staged again: no branch in the body)
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const int step = ch * KT + k;
      const int nstep = step + 1 < nsteps ? step + 1 : nsteps - 1;
      const int ks = (k + 2) % KT;                          // the tap that is sampled under this one
      const float* wsrc = k + 2 < KT ? cur : nxt;           // (taps 0, 1 of the next chunk: its window is complete since tap 5)
      const uint4* fcur = fragbase + (k % 3) * SP_OP;       // (9 taps per chunk: step % 3 == k % 3)
      const uint4* fnxt = fragbase + ((k + 1) % 3) * SP_OP;
      uint4* opw = opbuf + ((k + 2) % 3) * SP_OP + (wave * 3) * 64 + lane;
      const float4 tw = rw[ks];
      const float* wp_ = wsrc + half * 8 * CP + roff[ks];
      // (the first half batch of the NEXT tap's sampling is requested in slot 20 of this one: read at the top of its own tap, the
      // first combine waited for it right behind the barrier, with both waves of a SIMD in the same place)
      const float* wpn_ = (k + 3 < KT ? cur : nxt) + half * 8 * CP + roff[(k + 3) % KT];
      auto words = [&](const float* wp, int hb) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float* q = wp + (hb * 4 + c) * CP;
          raw[c][0] = q[0];
          raw[c][1] = q[WRP];
          raw[c][2] = q[1];
          raw[c][3] = q[WRP + 1];
        }
      };
      auto combine = [&](int c) {  // (one fma chain per value, kept scalar: see sp_split2)
        v[c] = __builtin_fmaf(tw.w, raw[c & 3][3], __builtin_fmaf(tw.z, raw[c & 3][2], __builtin_fmaf(tw.y, raw[c & 3][1], tw.x * raw[c & 3][0])));
        asm("" : "+v"(v[c]));
      };
      auto split_a = [&](int j) {
        q1[j] = sp_pack2(v[2 * j], v[2 * j + 1]);
        ra[j] = v[2 * j] - __builtin_bit_cast(float, q1[j] << 16);
        rb[j] = v[2 * j + 1] - __builtin_bit_cast(float, q1[j] & 0xffff0000u);
        asm("" : "+v"(ra[j]), "+v"(rb[j]));
      };
      auto split_b = [&](int j) {
        q2[j] = sp_pack2(ra[j], rb[j]);
        ra[j] = ra[j] - __builtin_bit_cast(float, q2[j] << 16);
        rb[j] = rb[j] - __builtin_bit_cast(float, q2[j] & 0xffff0000u);
        asm("" : "+v"(ra[j]), "+v"(rb[j]));
      };
      auto split_c = [&](int j) { q3[j] = sp_pack2(ra[j], rb[j]); };

      // top of the tap: weight fragments of the next tap (global), the first half batch of window words
#pragma unroll
      for (int p = 0; p < 3; ++p) anxt[p] = (wpa + (long long)nstep * MTW * 192)[(unsigned)(p * 64 + lane)];
      MODE_SB
      MODE_MF(2, b0, 0) combine(0); MODE_SB
      MODE_MF(2, b0, 1) combine(1); MODE_SB
      MODE_MF(2, b0, 2) combine(2); MODE_SB
      MODE_MF(2, b0, 3) combine(3); MODE_SB
      words(wp_, 1);
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) b1[gi] = fcur[(gi * 3 + 1) * 64];
      MODE_SB
      MODE_MF(1, b0, 0) split_a(0); MODE_SB
      MODE_MF(1, b0, 1) split_b(0); MODE_SB
      MODE_MF(1, b0, 2) split_c(0); split_a(1); MODE_SB
      MODE_MF(1, b0, 3) split_b(1); MODE_SB
      MODE_MF(0, b0, 0) split_c(1); combine(4); MODE_SB
      MODE_MF(0, b0, 1) combine(5); MODE_SB
      MODE_MF(0, b0, 2) combine(6); MODE_SB
      MODE_MF(0, b0, 3) combine(7); MODE_SB
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) b2[gi] = fcur[(gi * 3 + 2) * 64];
      MODE_SB
      MODE_MF(1, b1, 0) split_a(2); MODE_SB
      MODE_MF(1, b1, 1) split_b(2); MODE_SB
      MODE_MF(1, b1, 2) split_c(2); split_a(3); MODE_SB
      MODE_MF(1, b1, 3) split_b(3); MODE_SB
      MODE_MF(0, b1, 0) split_c(3); MODE_SB
      opw[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
      opw[64] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
      opw[128] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
      MODE_SB
      MODE_MF(0, b1, 1) MODE_SB
      // window of the next chunk: two phases loaded under each of taps 0..3, stored STAGE_LAG taps later (the whole window is in LDS
      // before tap 7 samples from it)
      if (k < 4) {
        issue(chn, 2 * k, 2 * k);
        issue(chn, 2 * k + 1, 2 * k + 1);
      }
      MODE_SB
      MODE_MF(0, b1, 2) MODE_SB
      MODE_MF(0, b1, 3) MODE_SB
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) b0n[gi] = fnxt[(gi * 3 + 0) * 64];
      words(wpn_, 0);
      MODE_SB
      MODE_MF(0, b2, 0) MODE_SB
      if (k >= STAGE_LAG && k < 4 + STAGE_LAG) commit(chn, 2 * (k - STAGE_LAG), 2 * (k - STAGE_LAG), nxt);
      MODE_SB
      MODE_MF(0, b2, 1) MODE_SB
      if (k >= STAGE_LAG && k < 4 + STAGE_LAG) commit(chn, 2 * (k - STAGE_LAG) + 1, 2 * (k - STAGE_LAG) + 1, nxt);
      MODE_SB
      MODE_MF(0, b2, 2) MODE_SB
      MODE_MF(0, b2, 3) MODE_SB
#pragma unroll
      for (int p = 0; p < 3; ++p) acur[p] = anxt[p];
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) b0[gi] = b0n[gi];
      sp_lds_barrier();
    }
  }
#undef MODE_MF
  }
#undef MODE_SB
  MODE_STAMP(2)

  // D[i = o][j = pixel of group gset + gi]
  const int hh = h0 + (wave / TW) * 32 + (lane & 31);
  const int cmax = d.Cog - mg * 128;
  // eval mode: the 16 folded-BatchNorm shifts of this lane's output channels, ONCE (the same for the four pixel groups): read inside the
  // store loop every store waited for a load of its own -- `shift` may alias y as far as the compiler knows (round 5, ISA scan: 256
  // `s_waitcnt vmcnt(0)` in this kernel)
  float shv[16];
  if (EPI) {
#pragma unroll
    for (int r = 0; r < 16; ++r) shv[r] = epi.shift[g * d.Cog + mg * 128 + min(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, cmax - 1)];
  }
  const bool full_m = m * 32 + 32 <= cmax;  // (uniform per wave) every channel of this M-tile exists: no test per store
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) {
    const int ww = w0 + gi;
    if (hh < d.H && ww < d.W) {
      float* yb = y + ((long long)b * d.Co + (long long)g * d.Cog + (long long)mg * 128) * HW + (long long)hh * d.sh + (long long)ww * d.sw;
      // eval mode: the residual values of this pixel group are requested together, ahead of its stores (`add` may alias `y` for all the
      // compiler knows: read next to the stores, each load waited for the store in front of it -- see deconv3d_kernel)
      float res[16];
      if (EPI) {
#pragma unroll
        for (int r = 0; r < 16; ++r) res[r] = 0.f;
        if (epi.add) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = min(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, cmax - 1);
            res[r] = epi.add[(yb - y) + (long long)co * HW];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      auto emit = [&](int r) {
        const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (EPI) {
          const float v = (acc[gi][r] + shv[r]) + res[r];
          yb[(long long)co * HW] = epi.relu ? relu_nan(v) : v;
        } else {
          yb[(long long)co * HW] = F16 ? acc[gi][r] * unscale : acc[gi][r];
        }
      };
      if (full_m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) emit(r);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < cmax) emit(r);
      }
      if (EPI) __builtin_amdgcn_sched_barrier(0);
    }
  }
  MODE_STAMP(3)
}

// =====================================================================================================================
// Weight gradient on the compact-window tiles with the split-bf16 arithmetic:  gW[o][c][k] = sum_p gy[o][p] * col[c][k][p],
// D[i = o][j = c] per tap on v_mfma_f32_32x32x16_bf16 with K = 16 PIXELS per instruction (sphere_bww_win_kernel: one pixel per fp32
// MFMA).  Same work items (half a plan tile = 32 rows x 4 columns of one sample, column by column), x window, record table, wave roles
// (wave v owns tap v for all four o-tiles; tap 8 is shared), partial-sum layout and reduction as sphere_bww_win_kernel.  New:
//   * gy is split when a column is staged: LDS holds [3 pieces][128 o][32 px] bf16 (row pitch 80 B: the 16 lanes of a b128 read
//     group hit 16 different 16-byte slots), so an A fragment (lane = o, 8 consecutive pixels) is ONE ds_read_b128 per piece;
//   * the B fragment (lane = c, 8 consecutive pixels) needs the sampled values pixel-fastest per channel, while sampling wants one
//     pixel per lane (its record in registers).  So a wave samples 16 pixels x 32 channels per K-step with lane = (pixel octet h,
//     channel octet q, pixel p8): 8 channels x (4 window words + 4 FMAs) for ITS pixel, transposes the 8 x 8 blocks through a
//     wave-private LDS scratch (8 ds_write_b32 + 8 ds_read_b32, conflict-free at pitch 17), and splits the 8 pixels of its channel
//     into the three bf16 fragments in registers -- no operand buffer, no extra barrier;
//   * per column a wave runs 2 K-steps of its own tap (2 x 24 MFMAs) and one (K-step, o-tile) piece of tap 8 (6 MFMAs): wave v takes
//     K-step v / 4, o-tile v % 4; the two shares of an o-tile are added through LDS at the end, in wave order.
// One accumulator per (tap, o-tile): a workgroup sums ~900 K-steps, i.e. ~5 400 roundings at the magnitude of the running sum
// against the fp32 kernel's ~14 000 (one per pixel); measured against float64 in tests/test_gpu_split.py.
constexpr int BS_GP = 20;                      // dwords per gy row: 16 (32 bf16) + 4 of padding
constexpr int BS_GYB = 3 * 128 * BS_GP;        // dwords of the gy column buffer (3 pieces)
constexpr int BS_SCRP = 17;                    // scratch row pitch (floats)
constexpr int BS_SCR = 32 * BS_SCRP;           // floats of one wave's transposition scratch
constexpr int BS_X2 = ((2 * BW_XW + 3) / 4) * 4;  // both x windows, rounded so that the gy buffer behind them is 16-byte aligned
constexpr int BS_LDS_DWORDS = BS_X2 + BS_GYB + 8 * BS_SCR;

// F16: the two-piece fp16 arithmetic of the other two spherical kernels (3v): gy is scaled by its tensor's power of two when a column is
// committed, the sampled operand through its four bilinear weights, the partial sums by the inverse of both; 12 MFMAs per K-step.
template <bool F16>
__global__ __launch_bounds__(NTHREADS) void sphere_bww_split_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                     float* __restrict__ part, WinDims d, const int4* __restrict__ tiles,
                                                                     const float4* __restrict__ rec_w, const int* __restrict__ rec_off,
                                                                     int ntiles, int S, const float* __restrict__ amax_g,
                                                                     const float* __restrict__ amax_x) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float sg = 1.f, sx = 1.f, unscale = 1.f;
  if (F16) {
    sg = sp_f16_scale_of(mode::absmax_load(amax_g));
    sx = sp_f16_scale_of(mode::absmax_load(amax_x));
    unscale = (1.f / sg) * (1.f / sx);
  }
  constexpr int NPC = F16 ? 2 : 3;  // pieces per value
  uint32_t* gyb = reinterpret_cast<uint32_t*>(smem + BS_X2);  // [3][128][BS_GP]
  constexpr int WRP = BW_WR, CP = BW_CP;
  const int s = blockIdx.x, cg = blockIdx.y;
  const int g = blockIdx.z / d.MG, mg = blockIdx.z % d.MG;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, half = lane >> 5;
  float* scr = smem + BS_X2 + BS_GYB + wave * BS_SCR;  // this wave's transposition scratch [32 c][BS_SCRP]
  const long long HW = (long long)d.H * d.W;
  const int T = ntiles * 2 * d.B;  // (sample, tile, half) items
  const int NCG = gridDim.y;

  f32x16 acc[4], acc8;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc[0][r] = acc[1][r] = acc[2][r] = acc[3][r] = 0.f;
    acc8[r] = 0.f;
  }
  for (int i = tid; i < BS_LDS_DWORDS; i += NTHREADS) smem[i] = 0.f;  // every word that may be read is finite

  const int omax = d.Cog - mg * 128;
  const int cmax = d.Cig - cg * BW_CG;
  // sampling role of this lane: pixel 16 ks + 8 sh + sp8 of the column, channels 8 sq .. 8 sq + 7 of the chunk
  const int sh = half, sq = j >> 3, sp8 = j & 7;
  const int ks8 = wave >> 2, m8 = wave & 3;  // this wave's piece of tap 8

  struct Item {
    int b, ti, hf, h0, w0, rbase, cbase;
  };
  auto item_of = [&](int t) {
    Item it;
    it.b = t / (ntiles * 2);
    const int r = t - it.b * ntiles * 2;
    it.ti = r >> 1;
    it.hf = r & 1;
    const int4 tl = tiles[it.ti];
    it.h0 = tl.x + it.hf * BW_TH;
    it.w0 = tl.y;
    it.rbase = (tl.z + it.hf * BW_TH) % d.H;
    it.cbase = tl.w & 0xffff;
    return it;
  };
  // x window staging exactly as in sphere_bww_win_kernel
  // (staging coordinates recomputed from an opaque copy of the thread index where they are used -- see issue_gy)
  float pxw[BW_NXW];
  bool pxw_ok = false;
  auto issue_xw = [&](const Item& it, int part) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    const int xcol = d.sh == 1 ? t_ >> 6 : t_ & (WC - 1), xrow = d.sh == 1 ? t_ & 63 : t_ >> 3;
    const bool xrow_ok = xrow < WRP;
    pxw_ok = xrow_ok && it.cbase + xcol < d.W;
    const int grow = (it.rbase + (xrow_ok ? xrow : 0)) % d.H;
    const float* xg = x + ((long long)it.b * d.Ci + (long long)g * d.Cig + (long long)cg * BW_CG) * HW +
                      (pxw_ok ? (long long)grow * d.sh + (long long)(it.cbase + xcol) * d.sw : 0);
#pragma unroll
    for (int c = 0; c < BW_NXW; ++c) pxw[c] = xg[(long long)min(part * BW_NXW + c, cmax - 1) * HW];
  };
  auto commit_xw = [&](float* xwdst, int part) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    const int xcol = d.sh == 1 ? t_ >> 6 : t_ & (WC - 1), xrow = d.sh == 1 ? t_ & 63 : t_ >> 3;
    if (xrow < WRP) {
      float* dst = xwdst + xcol * WRP + xrow;
#pragma unroll
      for (int c = 0; c < BW_NXW; ++c) dst[(part * BW_NXW + c) * BW_CP] = (pxw_ok && part * BW_NXW + c < cmax) ? pxw[c] : 0.f;
    }
  };
  // gy is requested TWO columns ahead (a column lasts ~1.5 us, less than a trip to HBM under load: with one column of lead the
  // commit between the two barriers waited for it -- measured 0.08 ms of a 0.36 ms call), into two register sets
  const int gpp_ = tid & 15, go0_ = tid >> 4;
  const int gpp = gpp_, go0 = go0_;
  struct GyPF {
    float g0[4], g1[4];
    unsigned ok;  // bit 2u: first pixel of pair real, bit 2u + 1: second
  };
  GyPF gpf[2];
  // records of the next column: own tap at K-steps 0 / 1, tap 8 at K-step ks8 -- for the pixel this lane samples
  float4 prw0, prw1, prw8;
  int pro0, pro1, pro8;
  auto issue_gy = [&](const Item& it, int wc, GyPF& pf) {
    // (gpp / go0 through an opaque copy: the row and the four clamped channel offsets derived from them are loop invariants, and kept
    // across the column loop they were the registers the allocator spilled -- every column began with two scratch reloads, each behind an
    // s_waitcnt vmcnt(0) that also drained the gy requested for the column after next.  Recomputed here they cost ~12 instructions.)
    int gpp = gpp_, go0 = go0_;
    asm volatile("" : "+v"(gpp), "+v"(go0));
    const int h = it.h0 + 2 * gpp, w = it.w0 + wc;
    const bool ok0 = h < d.H && w < d.W, ok1 = h + 1 < d.H && w < d.W;
    const float* gyb0 = gy + ((long long)it.b * d.Co + (long long)g * d.Cog + (long long)mg * 128) * HW +
                        (ok0 ? (long long)h * d.sh + (long long)w * d.sw : 0);
    const long long step1 = ok1 ? d.sh : 0;
    pf.ok = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = go0 + 32 * u;
      const float* p = gyb0 + (long long)min(o, omax - 1) * HW;
      pf.g0[u] = p[0];
      pf.g1[u] = p[step1];
      pf.ok |= ((ok0 && o < omax) ? 1u : 0u) << (2 * u) | ((ok1 && o < omax) ? 2u : 0u) << (2 * u);
    }
  };
  auto issue_rec = [&](const Item& it, int wc) {  // records: L2-resident, one column of lead is plenty
    const long long rcol = (((long long)it.ti * 2 + it.hf) * TW + wc) * BW_NREC;
    const int px0 = 8 * sh + sp8;
    prw0 = rec_w[rcol + wave * BW_TH + px0];
    pro0 = rec_off[rcol + wave * BW_TH + px0];
    prw1 = rec_w[rcol + wave * BW_TH + 16 + px0];
    pro1 = rec_off[rcol + wave * BW_TH + 16 + px0];
    prw8 = rec_w[rcol + 8 * BW_TH + 16 * ks8 + px0];
    pro8 = rec_off[rcol + 8 * BW_TH + 16 * ks8 + px0];
  };
  auto commit_col = [&](const GyPF& pf) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float a = (pf.ok >> (2 * u) & 1u) ? pf.g0[u] : 0.f, b2 = (pf.ok >> (2 * u + 1) & 1u) ? pf.g1[u] : 0.f;
      uint32_t p1, p2, p3;
      uint32_t* dst = gyb + (go0 + 32 * u) * BS_GP + gpp;
      if constexpr (F16) {
        sp_split2_f16(a * sg, b2 * sg, p1, p2);
      } else {
        sp_split2(a, b2, p1, p2, p3);
        dst[2 * 128 * BS_GP] = p3;
      }
      dst[0] = p1;
      dst[128 * BS_GP] = p2;
    }
  };
  // B fragment of one K-step: sample this lane's pixel for 8 channels, transpose through the scratch, split this lane's channel.
  // No wait between the scratch stores and loads: the LDS serves the instructions of one wave in order, and the compiler keeps
  // may-aliasing LDS accesses in program order -- an explicit s_waitcnt here would pin the whole sample in front of the MFMAs.
  auto sample = [&](const float* xw, const float4 w4_, const int ro, uint4 (&bf)[3]) {
    const float* xb = xw + (8 * sq) * CP + ro;
    float v[8], r_[8][4];
    // (all 32 window words requested before the first is used: as one loop the compiler issued every read right in front of its
    // use -- 16 serialised LDS round trips per sample, three samples per column; see sphere_fwd_split_kernel::sample_read)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float* q = xb + c * CP;
      r_[c][0] = q[0];
      r_[c][1] = q[WRP];
      r_[c][2] = q[1];
      r_[c][3] = q[WRP + 1];
    }
    __builtin_amdgcn_sched_barrier(0);
    // (F16: the operand's power-of-two scale rides on the four weights)
    const float4 w4 = F16 ? make_float4(w4_.x * sx, w4_.y * sx, w4_.z * sx, w4_.w * sx) : w4_;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      v[c] = __builtin_fmaf(w4.w, r_[c][3], __builtin_fmaf(w4.z, r_[c][2], __builtin_fmaf(w4.y, r_[c][1], w4.x * r_[c][0])));
      asm("" : "+v"(v[c]));
    }
    float* sw = scr + (8 * sq) * BS_SCRP + 8 * sh + sp8;
#pragma unroll
    for (int c = 0; c < 8; ++c) sw[c * BS_SCRP] = v[c];
    const float* sr = scr + j * BS_SCRP + 8 * half;
    float t8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t8[i] = sr[i];
    uint32_t q1[4], q2[4], q3[4];
    if constexpr (F16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sp_split2_f16(t8[2 * i], t8[2 * i + 1], q1[i], q2[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) sp_split2(t8[2 * i], t8[2 * i + 1], q1[i], q2[i], q3[i]);
      bf[2] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
    }
    bf[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    bf[1] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
  };
  // A fragments (gy, lane = o, 8 consecutive pixels) of the four o-tiles for K-step ks
  auto load_a = [&](int ks, uint4 (&a)[4][3]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const uint4* ga = reinterpret_cast<const uint4*>(gyb + (m * 32 + j) * BS_GP + 8 * ks + 4 * half);
#pragma unroll
      for (int p = 0; p < NPC; ++p) a[m][p] = ga[p * (128 * BS_GP / 4)];
    }
  };
  // the 24 (F16: 12) MFMAs of one K-step of this wave's tap: smallest terms first, consecutive MFMAs on different accumulators
  auto mma24 = [&](const uint4 (&a)[4][3], const uint4 (&bf)[3]) {
    if constexpr (F16) {
#define MODE_BS_TERM(PA, PB) _Pragma("unroll") for (int m = 0; m < 4; ++m) acc[m] = sp_mfma_f16(a[m][PA], bf[PB], acc[m]);
      MODE_BS_TERM(1, 0)
      MODE_BS_TERM(0, 1)
      MODE_BS_TERM(0, 0)
#undef MODE_BS_TERM
    } else {
#define MODE_BS_TERM(PA, PB) _Pragma("unroll") for (int m = 0; m < 4; ++m) acc[m] = sp_mfma(a[m][PA], bf[PB], acc[m]);
      MODE_BS_TERM(2, 0)
      MODE_BS_TERM(0, 2)
      MODE_BS_TERM(1, 1)
      MODE_BS_TERM(1, 0)
      MODE_BS_TERM(0, 1)
      MODE_BS_TERM(0, 0)
#undef MODE_BS_TERM
    }
  };
  // one MFMA, then up to 4 (F16: 7) vector-ALU instructions and 3 (F16: 4) LDS instructions, n times: spreads a sample (and the next
  // fragment reads) over the matrix instructions issued beside it -- half the MFMAs carry a sample that is three quarters of the work
#define MODE_BS_SPREAD(n)                                            \
  _Pragma("unroll") for (int i_ = 0; i_ < (F16 ? (n) / 2 : (n)); ++i_) { \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               \
    __builtin_amdgcn_sched_group_barrier(0x002, F16 ? 7 : 4, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x100, F16 ? 4 : 3, 0);     \
  }                                                                  \
  __builtin_amdgcn_sched_barrier(0);

  Item cur = item_of(s);
  __syncthreads();  // zero fill done
  for (int part = 0; part < TW; ++part) {
    issue_xw(cur, part);
    commit_xw(smem, part);
  }
  issue_gy(cur, 0, gpf[0]);
  issue_rec(cur, 0);
  commit_col(gpf[0]);
  issue_gy(cur, 1, gpf[1]);
  __syncthreads();

  int xbuf = 0;
  uint4 b0[3];
  sample(smem, prw0, pro0, b0);  // K-step 0 of the first column
  for (int t = s; t < T; t += S) {
    const bool more_items = t + S < T;
    const Item nxt = item_of(more_items ? t + S : t);
    const float* xw = smem + xbuf * BW_XW;
#pragma unroll
    for (int wc = 0; wc < TW; ++wc) {  // (unrolled: the gy register set of a column is wc % 2)
      const bool last_col = wc == TW - 1;
      const bool have_next = !last_col || more_items;
      const float4 rw1 = prw1, rw8 = prw8;
      const int ro1 = pro1, ro8 = pro8;
      // Requests in the order they are consumed (vmcnt retires in order): the next item's window part (committed at the end of this
      // column), the next column's records, and last the gy of the column after next (its register set held this column's gy, committed
      // at the end of the previous column) -- requested first, the window commit's wait drained it too.  All unconditional (`nxt` is
      // the current item again when there is no next one): a request under a branch turns the waits behind it into vmcnt(0).
      issue_xw(nxt, wc);
      issue_rec(last_col ? nxt : cur, last_col ? 0 : wc + 1);
      issue_gy(wc < 2 ? cur : nxt, wc < 2 ? wc + 2 : wc - 2, gpf[wc & 1]);
      __builtin_amdgcn_sched_barrier(0);

      uint4 a[4][3], b1[3], b8[3];
      // K-step 0 of the own tap beside the sampling of K-step 1
      load_a(0, a);
      sample(xw, rw1, ro1, b1);
      mma24(a, b0);
      MODE_BS_SPREAD(24)
      // K-step 1 beside the sampling of this wave's piece of tap 8
      load_a(1, a);
      sample(xw, rw8, ro8, b8);
      mma24(a, b1);
      MODE_BS_SPREAD(24)
      // tap 8: o-tile m8, K-step ks8
      {
        const uint4* ga = reinterpret_cast<const uint4*>(gyb + (m8 * 32 + j) * BS_GP + 8 * ks8 + 4 * half);
        const uint4 a1 = ga[0], a2 = ga[128 * BS_GP / 4];
        if constexpr (F16) {
          acc8 = sp_mfma_f16(a2, b8[0], acc8);
          acc8 = sp_mfma_f16(a1, b8[1], acc8);
          acc8 = sp_mfma_f16(a1, b8[0], acc8);
        } else {
          const uint4 a3 = ga[2 * (128 * BS_GP / 4)];
          acc8 = sp_mfma(a3, b8[0], acc8);
          acc8 = sp_mfma(a1, b8[2], acc8);
          acc8 = sp_mfma(a2, b8[1], acc8);
          acc8 = sp_mfma(a2, b8[0], acc8);
          acc8 = sp_mfma(a1, b8[1], acc8);
          acc8 = sp_mfma(a1, b8[0], acc8);
        }
      }
      if (more_items) commit_xw(smem + (xbuf ^ 1) * BW_XW, wc);  // the other window buffer: nobody reads it now

      __syncthreads();  // everyone is done with this column's gy
      if (have_next) commit_col(gpf[(wc + 1) & 1]);
      __syncthreads();
      // K-step 0 of the next column (after the barrier: at an item boundary its window was completed by the commit above)
      if (have_next) sample(last_col ? smem + (xbuf ^ 1) * BW_XW : xw, prw0, pro0, b0);
    }
    cur = nxt;
    xbuf ^= 1;
  }
#undef MODE_BS_SPREAD

  // partials: [tap][128 o][32 c] per (slice, z, channel group), the layout reduce_gw_win sums
  float* pb = part + ((((long long)s * gridDim.z + blockIdx.z) * NCG + cg) * BW_SLOTS) * (128 * BW_CG);
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      pb[((long long)wave * 128 + o) * BW_CG + j] = F16 ? acc[m][r] * unscale : acc[m][r];
    }
  // tap 8: o-tile m8 = the share of K-step 0 (waves 0..3) + the share of K-step 1 (waves 4..7), added in that order through LDS
  float* red = smem;  // [128 o][32 c]; the item loop ended with a barrier
  if (wave < 4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(m8 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * BW_CG + j] = F16 ? acc8[r] * unscale : acc8[r];
  }
  __syncthreads();
  if (wave >= 4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(m8 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * BW_CG + j] += F16 ? acc8[r] * unscale : acc8[r];
  }
  __syncthreads();
  for (int i = tid; i < 128 * BW_CG; i += NTHREADS) pb[(long long)8 * 128 * BW_CG + i] = red[i];
}

// The same contraction for the polar items (sphere_bww_polar_kernel's work items: one column of 32 pixels with nine per-tap windows,
// staged in place per item), on the split-bf16 arithmetic: the sampling / transposition / MFMA sequence of sphere_bww_split_kernel on
// the per-tap window layout.
constexpr int BQ_X = ((BP_XW + 3) / 4) * 4;
constexpr int BQ_LDS_DWORDS = BQ_X + BS_GYB + 8 * BS_SCR;

__global__ __launch_bounds__(NTHREADS) void sphere_bww_polar_split_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                           float* __restrict__ part, WinDims d,
                                                                           const int* __restrict__ pitems, const float4* __restrict__ rec_w,
                                                                           const int* __restrict__ rec_off, int nitems, int S, int s_base) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xw = smem;
  uint32_t* gyb = reinterpret_cast<uint32_t*>(smem + BQ_X);
  constexpr int WRP = BP_WR, CP = BP_CP;
  const int s = blockIdx.x, cg = blockIdx.y;
  const int g = blockIdx.z / d.MG, mg = blockIdx.z % d.MG;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, half = lane >> 5;
  float* scr = smem + BQ_X + BS_GYB + wave * BS_SCR;
  const long long HW = (long long)d.H * d.W;
  const int T = nitems * d.B;
  const int omax = d.Cog - mg * 128, cmax = d.Cig - cg * BW_CG;
  const int sh = half, sq = j >> 3, sp8 = j & 7;
  const int ks8 = wave >> 2, m8 = wave & 3;
  const int gpp = tid & 15, go0 = tid >> 4;

  f32x16 acc[4], acc8;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc[0][r] = acc[1][r] = acc[2][r] = acc[3][r] = 0.f;
    acc8[r] = 0.f;
  }
  for (int i = tid; i < BQ_LDS_DWORDS; i += NTHREADS) smem[i] = 0.f;

  constexpr int NE = KT * BP_TAPW;  // 612 window words per channel
  const int e0 = tid, e1 = tid + NTHREADS;
  const bool has1 = e1 < NE;
  const int k0 = e0 / BP_TAPW, k1 = (has1 ? e1 : 0) / BP_TAPW;
  const int col0 = (e0 % BP_TAPW) / BP_WR, r0 = e0 % BP_WR;
  const int col1 = ((has1 ? e1 : 0) % BP_TAPW) / BP_WR, r1 = (has1 ? e1 : 0) % BP_WR;

  auto sample = [&](const float4 w4, const int ro, uint4 (&bf)[3]) {
    const float* xb = xw + (8 * sq) * CP + ro;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float* q = xb + c * CP;
      v[c] = __builtin_fmaf(w4.w, q[WRP + 1], __builtin_fmaf(w4.z, q[1], __builtin_fmaf(w4.y, q[WRP], w4.x * q[0])));
      asm("" : "+v"(v[c]));
    }
    float* sw = scr + (8 * sq) * BS_SCRP + 8 * sh + sp8;
#pragma unroll
    for (int c = 0; c < 8; ++c) sw[c * BS_SCRP] = v[c];
    const float* sr = scr + j * BS_SCRP + 8 * half;
    float t8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t8[i] = sr[i];
    uint32_t q1[4], q2[4], q3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sp_split2(t8[2 * i], t8[2 * i + 1], q1[i], q2[i], q3[i]);
    bf[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    bf[1] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
    bf[2] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
  };
  auto load_a = [&](int ks, uint4 (&a)[4][3]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const uint4* ga = reinterpret_cast<const uint4*>(gyb + (m * 32 + j) * BS_GP + 8 * ks + 4 * half);
#pragma unroll
      for (int p = 0; p < 3; ++p) a[m][p] = ga[p * (128 * BS_GP / 4)];
    }
  };
  auto mma24 = [&](const uint4 (&a)[4][3], const uint4 (&bf)[3]) {
#define MODE_BS_TERM(PA, PB) _Pragma("unroll") for (int m = 0; m < 4; ++m) acc[m] = sp_mfma(a[m][PA], bf[PB], acc[m]);
    MODE_BS_TERM(2, 0)
    MODE_BS_TERM(0, 2)
    MODE_BS_TERM(1, 1)
    MODE_BS_TERM(1, 0)
    MODE_BS_TERM(0, 1)
    MODE_BS_TERM(0, 0)
#undef MODE_BS_TERM
  };

  for (int t = s; t < T; t += S) {
    const int b = t / nitems, it = t - b * nitems;
    const int* pi = pitems + (long long)it * BP_ITEM_INTS;
    const int h0 = pi[0], w = pi[1];
    __syncthreads();  // previous item consumed (first pass: zero fill done)
    {
      const int rb0 = pi[2 + k0], cb0 = pi[2 + KT + k0], rb1 = pi[2 + k1], cb1 = pi[2 + KT + k1];
      const bool ok0 = cb0 + col0 < d.W, ok1 = has1 && cb1 + col1 < d.W;
      const float* xg = x + ((long long)b * d.Ci + (long long)g * d.Cig + (long long)cg * BW_CG) * HW;
      const long long a0 = ok0 ? (long long)((rb0 + r0) % d.H) * d.sh + (long long)(cb0 + col0) * d.sw : 0;
      const long long a1 = ok1 ? (long long)((rb1 + r1) % d.H) * d.sh + (long long)(cb1 + col1) * d.sw : 0;
#pragma unroll 1
      for (int c8 = 0; c8 < BW_CG; c8 += 8) {
        float v0[8], v1[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const long long co = (long long)min(c8 + c, cmax - 1) * HW;
          v0[c] = xg[co + a0];
          v1[c] = xg[co + a1];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          xw[(c8 + c) * BP_CP + e0] = (ok0 && c8 + c < cmax) ? v0[c] : 0.f;
          if (has1) xw[(c8 + c) * BP_CP + e1] = (ok1 && c8 + c < cmax) ? v1[c] : 0.f;
        }
      }
    }
    {
      const int h = h0 + 2 * gpp;
      const bool ok0 = h < d.H && w < d.W, ok1 = h + 1 < d.H && w < d.W;
      const float* gyb0 = gy + ((long long)b * d.Co + (long long)g * d.Cog + (long long)mg * 128) * HW +
                          (ok0 ? (long long)h * d.sh + (long long)w * d.sw : 0);
      const long long step1 = ok1 ? d.sh : 0;
      float v0[4], v1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* p = gyb0 + (long long)min(go0 + 32 * u, omax - 1) * HW;
        v0[u] = p[0];
        v1[u] = p[step1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool oo = go0 + 32 * u < omax;
        uint32_t p1, p2, p3;
        sp_split2((ok0 && oo) ? v0[u] : 0.f, (ok1 && oo) ? v1[u] : 0.f, p1, p2, p3);
        uint32_t* dst = gyb + (go0 + 32 * u) * BS_GP + gpp;
        dst[0] = p1;
        dst[128 * BS_GP] = p2;
        dst[2 * 128 * BS_GP] = p3;
      }
    }
    const long long rbase = (long long)it * BW_NREC;
    const int px0 = 8 * sh + sp8;
    const float4 rw0 = rec_w[rbase + wave * BW_TH + px0], rw1 = rec_w[rbase + wave * BW_TH + 16 + px0];
    const float4 rw8 = rec_w[rbase + 8 * BW_TH + 16 * ks8 + px0];
    const int ro0 = rec_off[rbase + wave * BW_TH + px0], ro1 = rec_off[rbase + wave * BW_TH + 16 + px0];
    const int ro8 = rec_off[rbase + 8 * BW_TH + 16 * ks8 + px0];
    __syncthreads();
    uint4 a[4][3], b0[3], b1[3], b8[3];
    sample(rw0, ro0, b0);
    load_a(0, a);
    sample(rw1, ro1, b1);
    mma24(a, b0);
    load_a(1, a);
    sample(rw8, ro8, b8);
    mma24(a, b1);
    {
      const uint4* ga = reinterpret_cast<const uint4*>(gyb + (m8 * 32 + j) * BS_GP + 8 * ks8 + 4 * half);
      const uint4 a1 = ga[0], a2 = ga[128 * BS_GP / 4], a3 = ga[2 * (128 * BS_GP / 4)];
      acc8 = sp_mfma(a3, b8[0], acc8);
      acc8 = sp_mfma(a1, b8[2], acc8);
      acc8 = sp_mfma(a2, b8[1], acc8);
      acc8 = sp_mfma(a2, b8[0], acc8);
      acc8 = sp_mfma(a1, b8[1], acc8);
      acc8 = sp_mfma(a1, b8[0], acc8);
    }
  }
  __syncthreads();
  float* pb = part + ((((long long)(s_base + s) * gridDim.z + blockIdx.z) * gridDim.y + cg) * BW_SLOTS) * (128 * BW_CG);
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      pb[((long long)wave * 128 + o) * BW_CG + j] = acc[m][r];
    }
  float* red = smem;
  if (wave < 4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(m8 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * BW_CG + j] = acc8[r];
  }
  __syncthreads();
  if (wave >= 4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(m8 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * BW_CG + j] += acc8[r];
  }
  __syncthreads();
  for (int i = tid; i < 128 * BW_CG; i += NTHREADS) pb[(long long)8 * 128 * BW_CG + i] = red[i];
}

size_t win_lds_bytes(int wr, bool pipe) { return ((size_t)(pipe ? 2 : 1) * CCH * chan_pitch(wr) + wr + 8) * sizeof(float); }
// the tall tiles of the split forward keep two pair steps of weight fragments (2 x 12 KB) behind their windows
constexpr size_t TALL_W_BYTES = 2 * (size_t)MTW * 3 * 64 * sizeof(uint4);
size_t tall_split_lds_bytes(int wr) { return ((win_lds_bytes(wr, true) + 15) / 16) * 16 + TALL_W_BYTES; }
bool wrap_is_pipelined(int H) { return H + 1 <= WR_PIPE_MAX && win_lds_bytes(H + 1, true) <= 160 * 1024; }

// out[p][w][h] = in[p][h][w] for P planes of H x W: 32x32 tiles through LDS, both sides coalesced
__global__ __launch_bounds__(256) void transpose_planes_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W) {
  __shared__ float tile[32][33];
  const long long plane = (long long)blockIdx.z * H * W;
  const int w0 = blockIdx.x * 32, h0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int h = h0 + ty + i, w = w0 + tx;
    if (h < H && w < W) tile[ty + i][tx] = in[plane + (long long)h * W + w];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int w = w0 + ty + i, h = h0 + tx;
    if (h < H && w < W) out[plane + (long long)w * H + h] = tile[tx][ty + i];
  }
}

int make_win_dims(WinDims& d, int B, int Ci, int H, int W, int Co, int Kh, int Kw, int groups, const char* who) {
  MODE_REQUIRE(B >= 0 && Ci > 0 && H > 0 && W > 0 && Co > 0 && groups > 0, MODE_ERR_BAD_ARG, "%s: non-positive size", who);
  MODE_REQUIRE(Kh * Kw == KT, MODE_ERR_UNSUPPORTED, "%s: the windowed kernels are built for %d taps (got %dx%d)", who, KT, Kh, Kw);
  MODE_REQUIRE(Ci % groups == 0 && Co % groups == 0, MODE_ERR_BAD_ARG, "%s: channels not divisible by groups", who);
  MODE_REQUIRE((long long)H * W < (1LL << 30), MODE_ERR_UNSUPPORTED, "%s: image too large", who);
  d.B = B; d.Ci = Ci; d.H = H; d.W = W; d.Co = Co; d.G = groups;
  d.Cig = Ci / groups;
  d.Cog = Co / groups;
  d.NCH = mode::cdiv(d.Cig, CCH);
  d.MG = mode::cdiv(d.Cog, 128);
  d.wr = H + 1;
  d.sh = W;
  d.sw = 1;
  d.wrap_pipe = wrap_is_pipelined(H) ? 1 : 0;
  d.accumulate = 0;
  return MODE_OK;
}


}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Host-side tile plan.  tiles_host[4*i .. 4*i+3] = (h0, w0, rbase, cbase) ordered by class; counts[c] = tiles of class c:
//   0: window of 81 rows   1: 145 rows   2: all H rows + 1 (wraps around)   3: does not fit (caller must use the general path)
// and the class is also stored in bits 16.. of the 4th word (cbase | class << 16).
// Inside a class the tiles are ordered so that, with the round-robin workgroup -> XCD assignment, the tiles that share rows of
// the output (and cache lines of the input window) run on the same XCD and meet in its L2.
extern "C" size_t mode_sphere_plan_max_tiles(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  return (size_t)mode::cdiv(H, TH) * mode::cdiv(W, TW);
}

extern "C" int mode_sphere_plan_build(const float* pos_host, int H, int W, int Kh, int Kw, int32_t* tiles_host, int32_t* counts) {
  MODE_REQUIRE(pos_host && tiles_host && counts, MODE_ERR_BAD_ARG, "mode_sphere_plan_build: null pointer");
  MODE_REQUIRE(H > 0 && W > 0 && Kh > 0 && Kw > 0, MODE_ERR_BAD_ARG, "mode_sphere_plan_build: non-positive size");
  const int KK = Kh * Kw;
  const long long HW = (long long)H * W;
  const int nth = mode::cdiv(H, TH), ntw = mode::cdiv(W, TW);
  std::vector<int32_t> cls[4];
  // order: groups of 8 row-blocks; inside a group all column blocks; inside a column block the 8 row-blocks -> index % 8
  // (the XCD) is the row-block, for every column block
  for (int hg = 0; hg < nth; hg += kNumXCD)
    for (int tw = 0; tw < ntw; ++tw)
      for (int hs = 0; hs < kNumXCD && hg + hs < nth; ++hs) {
        const int h0 = (hg + hs) * TH, w0 = tw * TW;
        int dmin = 1 << 30, dmax = -(1 << 30), cmin = 1 << 30, cmax = -(1 << 30);
        bool any = false;
        for (int k = 0; k < KK; ++k)
          for (int h = h0; h < std::min(h0 + TH, H); ++h)
            for (int w = w0; w < std::min(w0 + TW, W); ++w) {
              int r0, c0;
              float4 wt;
              const long long idx = (long long)h * W + w;
              if (!mode::tap_record_fixed(pos_host[(2 * k) * HW + idx], pos_host[(2 * k + 1) * HW + idx], H, W, r0, c0, wt)) continue;
              if (wt.x == 0.f && wt.y == 0.f && wt.z == 0.f && wt.w == 0.f) continue;
              int dr = r0 - h0;  // wrapped into (-H/2, H/2]
              dr %= H;
              if (dr > H / 2) dr -= H;
              if (dr <= -(H + 1) / 2) dr += H;
              dmin = std::min(dmin, dr);
              dmax = std::max(dmax, dr);
              cmin = std::min(cmin, c0);
              cmax = std::max(cmax, c0);
              any = true;
            }
        int c = 0, rbase = h0, cbase = std::min(w0, std::max(W - WC, 0));
        if (any) {
          const int rows = dmax - dmin + 2;  // + the second corner row
          const int cols = cmax - cmin + 2;
          cbase = cmin;
          rbase = ((h0 + dmin) % H + H) % H;
          if (cols > WC) {
            c = 3;
          } else if (rows <= WR_SMALL && halves_fit(pos_host, H, W, KK, h0, w0, rbase, cbase)) {
            c = 0;
          } else if (rows <= WR_MID) {
            c = 1;
          } else {
            c = 2;  // whole axis: any start works, take 0 so that no row index wraps twice
            rbase = 0;
            if (win_lds_bytes(H + 1, false) > 160 * 1024) c = 3;
          }
        }
        if (cbase >= (1 << 16)) c = 3;
        cls[c].insert(cls[c].end(), {h0, w0, rbase, cbase | (c << 16)});
      }
  size_t o = 0;
  for (int c = 0; c < 4; ++c) counts[c] = (int32_t)(cls[c].size() / 4);
  for (int c : {2, 1, 0, 3}) {  // tall windows first: they are the slowest tiles of the launch
    if (!cls[c].empty()) std::memcpy(tiles_host + o, cls[c].data(), cls[c].size() * sizeof(int32_t));
    o += cls[c].size();
  }
  return MODE_OK;
}

extern "C" size_t mode_sphere_conv_win_wpack_bytes(int Ci, int Co, int Kh, int Kw, int groups) {
  if (Ci <= 0 || Co <= 0 || groups <= 0 || Kh * Kw != KT || Ci % groups || Co % groups) return 0;
  const int Cig = Ci / groups, Cog = Co / groups;
  const size_t f32 = (size_t)groups * mode::cdiv(Cog, 128) * mode::cdiv(Cig, CCH) * KT * MTW * 64 * 4 + (size_t)Co;  // + shifts
  const size_t split = (size_t)groups * mode::cdiv(Cog, 128) * mode::cdiv(Cig, SP_CCH) * KT * MTW * 3 * 64 * 4;       // split-bf16 fragments
  const size_t tall = (size_t)groups * mode::cdiv(Cog, 128) * mode::cdiv(Cig, CCH) * TP * MTW * 3 * 64 * 4;            // ... of the tall tiles
  return (((f32 + 3) / 4) * 4 + split + tall) * sizeof(float);
}

namespace {
template <bool EPI>
int fwd_win_launch(const float* x, const float* pos, float* y, const float* wpack, const int32_t* tiles, int n_small, int n_mid,
                   int n_wrap, int B, const WinDims& d, hipStream_t st, const Epi& epi);
}

static int sphere_conv_fwd_win_impl(const float* x, const float* pos, const float* w, float* y, float* wpack, const int32_t* tiles,
                                    int n_small, int n_mid, int n_wrap, int B, int Ci, int H, int W, int Co, int Kh, int Kw, int groups,
                                    int transposed, mode_stream_t stream, const mode_bn_epilogue* bn, int split = 0,
                                    const float* amax_x = nullptr, const float* amax_w = nullptr) {
  WinDims d;
  int rc = make_win_dims(d, B, Ci, H, W, Co, Kh, Kw, groups, "mode_sphere_conv_fwd_win");
  MODE_REQUIRE((amax_x == nullptr) == (amax_w == nullptr) && !(amax_x && bn), MODE_ERR_BAD_ARG,
               "mode_sphere_conv_fwd_win: the fp16 arithmetic takes both operand maxima and no BatchNorm epilogue");
  const bool f16 = amax_x != nullptr;
  if (rc != MODE_OK) return rc;
  if (transposed) {
    d.sh = 1;
    d.sw = H;
  }
  MODE_REQUIRE(n_small >= 0 && n_mid >= 0 && n_wrap >= 0 && (size_t)n_small + n_mid + n_wrap == mode_sphere_plan_max_tiles(H, W),
               MODE_ERR_BAD_ARG, "mode_sphere_conv_fwd_win: the plan must cover every tile (%d + %d + %d given, %zu tiles)", n_small, n_mid,
               n_wrap, mode_sphere_plan_max_tiles(H, W));
  if (B == 0) return MODE_OK;
  MODE_REQUIRE(x && pos && w && y && wpack && tiles, MODE_ERR_BAD_ARG, "mode_sphere_conv_fwd_win: null pointer");
  MODE_REQUIRE(B <= 65535 && d.G * d.MG <= 65535, MODE_ERR_UNSUPPORTED, "mode_sphere_conv_fwd_win: grid limit");
  hipStream_t st = mode::as_stream(stream);
  const long long npack = (long long)d.G * d.MG * d.NCH * KT * MTW * 64 * 4;
  const bool split_path = split && n_small > 0 && d.Cig % SP_CCH == 0;
  const bool wrap_on_fp32 = n_wrap > 0 && (!d.wrap_pipe || tall_split_lds_bytes(d.wr) > 160 * 1024);
  // the fp32 fragment layout (and the folded-BatchNorm shifts behind it) is read by the fp32 kernels and by the eval epilogue: a training
  // call whose tiles all run on the split kernel does not need it
  if (mode::pack_needed() && (!split_path || bn || wrap_on_fp32))
    hipLaunchKernelGGL(pack_w_win, dim3(mode::cdiv(npack, 256)), dim3(256), 0, st, w, wpack, d, bn ? 1 : 0, bn ? *bn : mode_bn_epilogue());
  const Epi epi = make_epi(bn, wpack + npack);
  if (split_path) {
    // small-window tiles (the tail of the tile list) on the split-bf16 kernel, the tall-window classes on the fp32 kernels
    const int NCH16 = d.Cig / SP_CCH;
    uint4* wps = reinterpret_cast<uint4*>(wpack + ((npack + d.Co + 3) / 4) * 4);
    const long long nsplit = (long long)d.G * d.MG * NCH16 * KT * MTW * 64;
    if (mode::pack_needed()) {
      if (f16)
        hipLaunchKernelGGL(pack_w_win_split<true>, dim3(mode::cdiv(nsplit, 256)), dim3(256), 0, st, w, wps, d, NCH16, 0, mode_bn_epilogue(), amax_w);
      else
        hipLaunchKernelGGL(pack_w_win_split<false>, dim3(mode::cdiv(nsplit, 256)), dim3(256), 0, st, w, wps, d, NCH16, bn ? 1 : 0,
                           bn ? *bn : mode_bn_epilogue(), (const float*)nullptr);
    }
    uint4* wpt = wps + nsplit * 3;  // fragments of the tall-window tiles: K = 8 channels x 2 taps
    if (n_mid + n_wrap > 0) {
      const long long ntall = (long long)d.G * d.MG * d.NCH * TP * MTW * 64;
      if (mode::pack_needed()) {
        if (f16)
          hipLaunchKernelGGL(pack_w_win_split_tall<true>, dim3(mode::cdiv(ntall, 256)), dim3(256), 0, st, w, wpt, d, 0, mode_bn_epilogue(), amax_w);
        else
          hipLaunchKernelGGL(pack_w_win_split_tall<false>, dim3(mode::cdiv(ntall, 256)), dim3(256), 0, st, w, wpt, d, bn ? 1 : 0,
                             bn ? *bn : mode_bn_epilogue(), (const float*)nullptr);
      }
    }
    // one launch for all tiles; wrap-around tiles that cannot be double-buffered keep their own kernel
    const int4* tl = reinterpret_cast<const int4*>(tiles);
    int n_all = n_small + n_mid + n_wrap;
    if (wrap_on_fp32) {
      rc = bn ? fwd_win_launch<true>(x, pos, y, wpack, tiles, 0, 0, n_wrap, B, d, st, epi)
              : fwd_win_launch<false>(x, pos, y, wpack, tiles, 0, 0, n_wrap, B, d, st, epi);
      if (rc != MODE_OK) return rc;
      tl += n_wrap;
      n_all -= n_wrap;
      n_wrap = 0;
    }
    size_t lds = SP3_LDS_BYTES;
    if (n_mid > 0) lds = std::max(lds, tall_split_lds_bytes(WR_MID));
    if (n_wrap > 0) lds = std::max(lds, tall_split_lds_bytes(d.wr));
    if (bn) {
      rc = mode::allow_lds(sphere_fwd_split_kernel<true>, lds, "mode_sphere_conv_fwd_win_split");
      if (rc != MODE_OK) return rc;
      hipLaunchKernelGGL(sphere_fwd_split_kernel<true>, dim3(n_all, B, d.G * d.MG), dim3(NTHREADS), lds, st, x, pos, wps, wpt, y, d, NCH16, tl,
                         epi, amax_x, amax_w);
    } else if (f16) {
      rc = mode::allow_lds(sphere_fwd_split_kernel<false, true>, lds, "mode_sphere_conv_fwd_win_split");
      if (rc != MODE_OK) return rc;
      hipLaunchKernelGGL((sphere_fwd_split_kernel<false, true>), dim3(n_all, B, d.G * d.MG), dim3(NTHREADS), lds, st, x, pos, wps, wpt, y, d, NCH16,
                         tl, epi, amax_x, amax_w);
    } else {
      rc = mode::allow_lds(sphere_fwd_split_kernel<false>, lds, "mode_sphere_conv_fwd_win_split");
      if (rc != MODE_OK) return rc;
      hipLaunchKernelGGL(sphere_fwd_split_kernel<false>, dim3(n_all, B, d.G * d.MG), dim3(NTHREADS), lds, st, x, pos, wps, wpt, y, d, NCH16, tl,
                         epi, amax_x, amax_w);
    }
    return mode::check_launch("mode_sphere_conv_fwd_win_split");
  }
  return bn ? fwd_win_launch<true>(x, pos, y, wpack, tiles, n_small, n_mid, n_wrap, B, d, st, epi)
            : fwd_win_launch<false>(x, pos, y, wpack, tiles, n_small, n_mid, n_wrap, B, d, st, epi);
}

namespace {
template <bool EPI>
int fwd_win_launch(const float* x, const float* pos, float* y, const float* wpack, const int32_t* tiles, int n_small, int n_mid,
                   int n_wrap, int B, const WinDims& d, hipStream_t st, const Epi& epi) {
  // tile list order: wrap-around tiles, then mid, then small
  const float4* wp4 = reinterpret_cast<const float4*>(wpack);
  const int4* tl = reinterpret_cast<const int4*>(tiles);
  int n_main = n_small + n_mid + n_wrap;
  int rc = MODE_OK;
  if (n_wrap > 0 && !d.wrap_pipe) {
    const size_t lds = win_lds_bytes(d.wr, false);
    rc = mode::allow_lds(sphere_fwd_win_tall_kernel<EPI>, lds, "mode_sphere_conv_fwd_win");
    if (rc != MODE_OK) return rc;
    hipLaunchKernelGGL(sphere_fwd_win_tall_kernel<EPI>, dim3(n_wrap, B, d.G * d.MG), dim3(NTHREADS), lds, st, x, pos, wp4, y, d, tl, epi);
    tl += n_wrap;
    n_main -= n_wrap;
    n_wrap = 0;
  }
  if (n_main > 0) {
    size_t lds = 0;
    if (n_small > 0) lds = std::max(lds, win_lds_bytes(WR_SMALL, true));
    if (n_mid > 0) lds = std::max(lds, win_lds_bytes(WR_MID, true));
    if (n_wrap > 0) lds = std::max(lds, win_lds_bytes(d.wr, true));
    rc = mode::allow_lds(sphere_fwd_win_kernel<EPI>, lds, "mode_sphere_conv_fwd_win");
    if (rc != MODE_OK) return rc;
    hipLaunchKernelGGL(sphere_fwd_win_kernel<EPI>, dim3(n_main, B, d.G * d.MG), dim3(NTHREADS), lds, st, x, pos, wp4, y, d, tl, epi);
  }
  return mode::check_launch("mode_sphere_conv_fwd_win");
}
}  // namespace

extern "C" int mode_sphere_conv_fwd_win(const float* x, const float* pos, const float* w, float* y, float* wpack,
                                        const int32_t* tiles, int n_small, int n_mid, int n_wrap, int B, int Ci, int H, int W, int Co,
                                        int Kh, int Kw, int groups, int transposed, mode_stream_t stream) {
  return sphere_conv_fwd_win_impl(x, pos, w, y, wpack, tiles, n_small, n_mid, n_wrap, B, Ci, H, W, Co, Kh, Kw, groups, transposed, stream,
                                  nullptr);
}

// Small-window tiles on the split-bf16 matrix path (sphere_fwd_split_kernel), the other classes on the fp32 kernels; bn may be NULL.
extern "C" int mode_sphere_conv_fwd_win_split(const float* x, const float* pos, const float* w, const mode_bn_epilogue* bn, float* y,
                                              float* wpack, const int32_t* tiles, int n_small, int n_mid, int n_wrap, int B, int Ci, int H,
                                              int W, int Co, int Kh, int Kw, int groups, int transposed, mode_stream_t stream) {
  if (bn) {
    int rc = mode::check_bn(bn, "mode_sphere_conv_fwd_win_split");
    if (rc != MODE_OK) return rc;
  }
  return sphere_conv_fwd_win_impl(x, pos, w, y, wpack, tiles, n_small, n_mid, n_wrap, B, Ci, H, W, Co, Kh, Kw, groups, transposed, stream, bn,
                                  1);
}

// The training forward with the small-window tiles on the two-piece fp16 arithmetic: amax_x / amax_w = device scalars holding the largest
// finite magnitude of x and of w (mode_abs_max, or the BatchNorm pass that wrote x).  Same results contract as the split-bf16 entry.
extern "C" int mode_sphere_conv_fwd_win_split_f16(const float* x, const float* pos, const float* w, const float* amax_x, const float* amax_w,
                                                  float* y, float* wpack, const int32_t* tiles, int n_small, int n_mid, int n_wrap, int B,
                                                  int Ci, int H, int W, int Co, int Kh, int Kw, int groups, int transposed,
                                                  mode_stream_t stream) {
  MODE_REQUIRE(amax_x && amax_w, MODE_ERR_BAD_ARG, "mode_sphere_conv_fwd_win_split_f16: null maximum");
  return sphere_conv_fwd_win_impl(x, pos, w, y, wpack, tiles, n_small, n_mid, n_wrap, B, Ci, H, W, Co, Kh, Kw, groups, transposed, stream,
                                  nullptr, 1, amax_x, amax_w);
}

extern "C" int mode_sphere_conv_fwd_win_bn(const float* x, const float* pos, const float* w, const mode_bn_epilogue* bn, float* y,
                                           float* wpack, const int32_t* tiles, int n_small, int n_mid, int n_wrap, int B, int Ci, int H,
                                           int W, int Co, int Kh, int Kw, int groups, int transposed, mode_stream_t stream) {
  int rc = mode::check_bn(bn, "mode_sphere_conv_fwd_win_bn");
  if (rc != MODE_OK) return rc;
  return sphere_conv_fwd_win_impl(x, pos, w, y, wpack, tiles, n_small, n_mid, n_wrap, B, Ci, H, W, Co, Kh, Kw, groups, transposed, stream, bn);
}

// ---------------------------------------------------------------------------------------------------------------------
// Pixels of the tiles that are NOT of the small-window class, sorted by linear index (they go to the general kernels).
extern "C" int mode_sphere_plan_rest_pixels(const int32_t* tiles_host, const int32_t* counts, int H, int W, int32_t* pix_host,
                                            int32_t* n_pix) {
  MODE_REQUIRE(tiles_host && counts && pix_host && n_pix, MODE_ERR_BAD_ARG, "mode_sphere_plan_rest_pixels: null pointer");
  const int n = counts[0] + counts[1] + counts[2] + counts[3];
  std::vector<int32_t> v;
  for (int i = 0; i < n; ++i) {
    const int32_t* t = tiles_host + 4 * i;
    if ((t[3] >> 16) == 0) continue;
    for (int h = t[0]; h < std::min(t[0] + TH, H); ++h)
      for (int w = t[1]; w < std::min(t[1] + TW, W); ++w) v.push_back(h * W + w);
  }
  std::sort(v.begin(), v.end());
  if (!v.empty()) std::memcpy(pix_host, v.data(), v.size() * sizeof(int32_t));
  *n_pix = (int32_t)v.size();
  return MODE_OK;
}

// Sampling records of the small-window tiles for the weight-gradient kernel, in tile-list order:
//   index (((ti*2 + half)*4 + column)*9 + tap)*32 + row -> window offset (rec_off) and the 4 corner weights (rec_w, 4 floats)
extern "C" size_t mode_sphere_plan_records_count(int n_small) { return n_small > 0 ? (size_t)n_small * 2 * TW * BW_NREC : 0; }

extern "C" int mode_sphere_plan_records(const float* pos_host, const int32_t* tiles_host, const int32_t* counts, int H, int W,
                                        float* rec_w_host, int32_t* rec_off_host) {
  MODE_REQUIRE(pos_host && tiles_host && counts && rec_w_host && rec_off_host, MODE_ERR_BAD_ARG, "mode_sphere_plan_records: null pointer");
  const long long HW = (long long)H * W;
  const int32_t* small = tiles_host + 4 * (size_t)(counts[2] + counts[1]);  // list order: wrap-around, mid, small
  for (int ti = 0; ti < counts[0]; ++ti) {
    const int h0 = small[4 * ti], w0 = small[4 * ti + 1], rbase = small[4 * ti + 2], cbase = small[4 * ti + 3] & 0xffff;
    for (int hf = 0; hf < 2; ++hf) {
      const int rb = (rbase + hf * BW_TH) % H;
      for (int wc = 0; wc < TW; ++wc)
        for (int k = 0; k < KT; ++k)
          for (int px = 0; px < BW_TH; ++px) {
            const size_t o = ((((size_t)ti * 2 + hf) * TW + wc) * KT + k) * BW_TH + px;
            const int h = h0 + hf * BW_TH + px, w = w0 + wc;
            int r0 = 0, c0 = 0;
            float4 wt = make_float4(0.f, 0.f, 0.f, 0.f);
            bool live = false;
            if (h < H && w < W) {
              const long long idx = (long long)h * W + w;
              live = mode::tap_record_fixed(pos_host[(2 * k) * HW + idx], pos_host[(2 * k + 1) * HW + idx], H, W, r0, c0, wt);
            }
            live = live && !(wt.x == 0.f && wt.y == 0.f && wt.z == 0.f && wt.w == 0.f);
            rec_off_host[o] = live ? (c0 - cbase) * BW_WR + ((r0 - rb) % H + H) % H : 0;
            rec_w_host[4 * o + 0] = live ? wt.x : 0.f;
            rec_w_host[4 * o + 1] = live ? wt.y : 0.f;
            rec_w_host[4 * o + 2] = live ? wt.z : 0.f;
            rec_w_host[4 * o + 3] = live ? wt.w : 0.f;
          }
    }
  }
  return MODE_OK;
}

// Work items of the polar weight-gradient kernel: every tile that is NOT of the small-window class, split into 2 halves x 4
// columns.  pitems_host[20 * i ..] = (h0, w, rbase[9], cbase[9]); rec_w_host[4 * (i*288 + tap*32 + row)], rec_off_host[...] = the
// sampling records in the per-tap window layout.  Returns the number of items, or -1 (in *n_items) if some column does not fit
// its per-tap windows (34 rows x 2 columns per tap) -- the caller then keeps those pixels on the general kernel.
extern "C" size_t mode_sphere_plan_polar_max_items(const int32_t* counts) {
  return counts ? (size_t)(counts[1] + counts[2]) * 2 * TW : 0;
}

extern "C" int mode_sphere_plan_polar(const float* pos_host, const int32_t* tiles_host, const int32_t* counts, int H, int W,
                                      int32_t* pitems_host, float* rec_w_host, int32_t* rec_off_host, int32_t* n_items) {
  MODE_REQUIRE(pos_host && tiles_host && counts && pitems_host && rec_w_host && rec_off_host && n_items, MODE_ERR_BAD_ARG,
               "mode_sphere_plan_polar: null pointer");
  const long long HW = (long long)H * W;
  const int ntall = counts[2] + counts[1];  // list order: wrap-around, mid, small
  int ni = 0;
  for (int ti = 0; ti < ntall; ++ti) {
    const int th0 = tiles_host[4 * ti], tw0 = tiles_host[4 * ti + 1];
    for (int hf = 0; hf < 2; ++hf)
      for (int wc = 0; wc < TW; ++wc) {
        const int h0 = th0 + hf * BW_TH, w = tw0 + wc;
        if (h0 >= H || w >= W) continue;
        int32_t* pi = pitems_host + (size_t)ni * BP_ITEM_INTS;
        pi[0] = h0;
        pi[1] = w;
        for (int k = 0; k < KT; ++k) {
          // row SHIFT of every live pixel, n = r0 - (h0 + px) modulo H, taken relative to the first one so that a shift of about
          // half the axis (the far side of the sphere) does not straddle the wrap-around cut
          int nfirst = 0, nmin = 1 << 30, nmax = -(1 << 30), cmin = 1 << 30, cmax = -(1 << 30);
          bool have = false;
          for (int px = 0; px < BW_TH && h0 + px < H; ++px) {
            int r0, c0;
            float4 wt;
            const long long idx = (long long)(h0 + px) * W + w;
            if (!mode::tap_record_fixed(pos_host[(2 * k) * HW + idx], pos_host[(2 * k + 1) * HW + idx], H, W, r0, c0, wt)) continue;
            if (wt.x == 0.f && wt.y == 0.f && wt.z == 0.f && wt.w == 0.f) continue;
            int n = ((r0 - h0 - px) % H + H) % H;
            if (!have) {
              nfirst = n;
              have = true;
            }
            n = ((n - nfirst + H / 2) % H + H) % H - H / 2 + nfirst;  // within H/2 of the first shift
            nmin = std::min(nmin, n);
            nmax = std::max(nmax, n);
            cmin = std::min(cmin, c0);
            cmax = std::max(cmax, c0);
          }
          int rb = 0, cbs = 0;
          if (have) {
            // rows h0 + nmin .. h0 + 31 + nmax (+1 for the second corner) must fit the 34-row window, the columns its 2
            if (BW_TH - 1 + (nmax - nmin) + 2 > BP_WR || cmax - cmin + 2 > 2) {
              *n_items = -1;
              return MODE_OK;
            }
            rb = ((h0 + nmin) % H + H) % H;
            cbs = cmin;
          }
          pi[2 + k] = rb;
          pi[2 + KT + k] = cbs;
          for (int px = 0; px < BW_TH; ++px) {
            const size_t o = ((size_t)ni * KT + k) * BW_TH + px;
            int r0 = 0, c0 = 0;
            float4 wt = make_float4(0.f, 0.f, 0.f, 0.f);
            bool live = false;
            if (h0 + px < H) {
              const long long idx = (long long)(h0 + px) * W + w;
              live = mode::tap_record_fixed(pos_host[(2 * k) * HW + idx], pos_host[(2 * k + 1) * HW + idx], H, W, r0, c0, wt);
            }
            live = live && !(wt.x == 0.f && wt.y == 0.f && wt.z == 0.f && wt.w == 0.f);
            rec_off_host[o] = live ? k * BP_TAPW + (c0 - cbs) * BP_WR + ((r0 - rb) % H + H) % H : 0;
            rec_w_host[4 * o + 0] = live ? wt.x : 0.f;
            rec_w_host[4 * o + 1] = live ? wt.y : 0.f;
            rec_w_host[4 * o + 2] = live ? wt.z : 0.f;
            rec_w_host[4 * o + 3] = live ? wt.w : 0.f;
          }
        }
        ++ni;
      }
  }
  *n_items = ni;
  return MODE_OK;
}

int bww_polar_splits(const WinDims& d, int n_items) {
  const int T = n_items * d.B;
  const int wgs_per_slice = mode::cdiv(d.Cig, BW_CG) * d.G * d.MG;
  const int per = std::max(1, mode::cdiv((long long)T * wgs_per_slice, kNumCU));
  return std::max(1, mode::cdiv(T, per));
}

extern "C" size_t mode_sphere_conv_bwd_weight_win_workspace_bytes(int B, int Ci, int H, int W, int Co, int Kh, int Kw, int groups,
                                                                  int n_small, int n_rest_pixels, int n_polar_items) {
  WinDims d;
  if (make_win_dims(d, B, Ci, H, W, Co, Kh, Kw, groups, "mode_sphere_conv_bwd_weight_win_workspace_bytes") != MODE_OK) return 0;
  const size_t slice = (size_t)d.G * d.MG * mode::cdiv(d.Cig, BW_CG) * BW_SLOTS * 128 * BW_CG * sizeof(float);
  const size_t win = slice * (bww_win_splits(d, std::max(n_small, 1)) + bww_polar_splits(d, std::max(n_polar_items, 1)));
  return win + mode::sphere_bwd_weight_general_workspace(B, Ci, Co, Kh, Kw, H, W, groups, std::max(n_rest_pixels, 1));
}

// Weight gradient, ADDED to gw like mode_sphere_conv_bwd_weight: windowed kernel on the n_small compact tiles of the plan
// (tile list order: wrap-around, mid, small), the polar kernel on the n_polar_items column items of the other tiles
// (mode_sphere_plan_polar), or -- when those could not be planned -- the general kernels on their n_rest_pixels pixels.
static int bwd_weight_win_impl(const float* gy, const float* pos, const float* x, float* gw, float* workspace, const int32_t* tiles,
                               int n_small, int n_mid, int n_wrap, const float* rec_w, const int32_t* rec_off, const int32_t* rest_pixels,
                               int n_rest_pixels, const int32_t* pitems, const float* prec_w, const int32_t* prec_off, int n_polar_items,
                               int B, int Ci, int H, int W, int Co, int Kh, int Kw, int groups, const float* gy_t, const float* x_t,
                               mode_stream_t stream, int split, const float* amax_g = nullptr, const float* amax_x = nullptr);

extern "C" int mode_sphere_conv_bwd_weight_win(const float* gy, const float* pos, const float* x, float* gw, float* workspace,
                                               const int32_t* tiles, int n_small, int n_mid, int n_wrap, const float* rec_w,
                                               const int32_t* rec_off, const int32_t* rest_pixels, int n_rest_pixels,
                                               const int32_t* pitems, const float* prec_w, const int32_t* prec_off, int n_polar_items,
                                               int B, int Ci, int H, int W, int Co, int Kh, int Kw, int groups, const float* gy_t,
                                               const float* x_t, mode_stream_t stream) {
  return bwd_weight_win_impl(gy, pos, x, gw, workspace, tiles, n_small, n_mid, n_wrap, rec_w, rec_off, rest_pixels, n_rest_pixels, pitems,
                             prec_w, prec_off, n_polar_items, B, Ci, H, W, Co, Kh, Kw, groups, gy_t, x_t, stream, 0);
}

// The same with the compact-window tiles on the split-bf16 kernel (sphere_bww_split_kernel: fp32 operands split exactly into three
// bf16 pieces, six bf16 MFMAs per product, fp32 accumulation); polar items and listed pixels as above.
extern "C" int mode_sphere_conv_bwd_weight_win_split(const float* gy, const float* pos, const float* x, float* gw, float* workspace,
                                                     const int32_t* tiles, int n_small, int n_mid, int n_wrap, const float* rec_w,
                                                     const int32_t* rec_off, const int32_t* rest_pixels, int n_rest_pixels,
                                                     const int32_t* pitems, const float* prec_w, const int32_t* prec_off,
                                                     int n_polar_items, int B, int Ci, int H, int W, int Co, int Kh, int Kw, int groups,
                                                     const float* gy_t, const float* x_t, mode_stream_t stream) {
  return bwd_weight_win_impl(gy, pos, x, gw, workspace, tiles, n_small, n_mid, n_wrap, rec_w, rec_off, rest_pixels, n_rest_pixels, pitems,
                             prec_w, prec_off, n_polar_items, B, Ci, H, W, Co, Kh, Kw, groups, gy_t, x_t, stream, 1);
}

// The split call with the compact-window tiles on the two-piece fp16 arithmetic (3v): amax_g / amax_x = the maximum buffers of gy and x
// (the polar items keep three bf16 pieces; their partial sums go through the same reduction).
extern "C" int mode_sphere_conv_bwd_weight_win_split_f16(const float* gy, const float* pos, const float* x, const float* amax_g,
                                                         const float* amax_x, float* gw, float* workspace, const int32_t* tiles, int n_small,
                                                         int n_mid, int n_wrap, const float* rec_w, const int32_t* rec_off,
                                                         const int32_t* rest_pixels, int n_rest_pixels, const int32_t* pitems,
                                                         const float* prec_w, const int32_t* prec_off, int n_polar_items, int B, int Ci, int H,
                                                         int W, int Co, int Kh, int Kw, int groups, const float* gy_t, const float* x_t,
                                                         mode_stream_t stream) {
  MODE_REQUIRE(amax_g && amax_x, MODE_ERR_BAD_ARG, "mode_sphere_conv_bwd_weight_win_split_f16: null maximum");
  return bwd_weight_win_impl(gy, pos, x, gw, workspace, tiles, n_small, n_mid, n_wrap, rec_w, rec_off, rest_pixels, n_rest_pixels, pitems,
                             prec_w, prec_off, n_polar_items, B, Ci, H, W, Co, Kh, Kw, groups, gy_t, x_t, stream, 1, amax_g, amax_x);
}

static int bwd_weight_win_impl(const float* gy, const float* pos, const float* x, float* gw, float* workspace, const int32_t* tiles,
                               int n_small, int n_mid, int n_wrap, const float* rec_w, const int32_t* rec_off, const int32_t* rest_pixels,
                               int n_rest_pixels, const int32_t* pitems, const float* prec_w, const int32_t* prec_off, int n_polar_items,
                               int B, int Ci, int H, int W, int Co, int Kh, int Kw, int groups, const float* gy_t, const float* x_t,
                               mode_stream_t stream, int split, const float* amax_g, const float* amax_x) {
  WinDims d;
  int rc = make_win_dims(d, B, Ci, H, W, Co, Kh, Kw, groups, "mode_sphere_conv_bwd_weight_win");
  MODE_REQUIRE((amax_g == nullptr) == (amax_x == nullptr), MODE_ERR_BAD_ARG, "mode_sphere_conv_bwd_weight_win: both operand maxima, or neither");
  if (rc != MODE_OK) return rc;
  MODE_REQUIRE((gy_t == nullptr) == (x_t == nullptr), MODE_ERR_BAD_ARG, "mode_sphere_conv_bwd_weight_win: gy_t and x_t come together");
  if (gy_t) {
    d.sh = 1;
    d.sw = H;
  }
  MODE_REQUIRE(n_small >= 0 && n_mid >= 0 && n_wrap >= 0 && n_rest_pixels >= 0 && n_polar_items >= 0 &&
                   (size_t)n_small + n_mid + n_wrap == mode_sphere_plan_max_tiles(H, W),
               MODE_ERR_BAD_ARG, "mode_sphere_conv_bwd_weight_win: the plan must cover every tile");
  if (B == 0) return MODE_OK;
  // (the NCHW tensors are only read by the pixel-list fallback: with the transposed copies and no listed pixel they may be null)
  MODE_REQUIRE(((gy && x) || (gy_t && n_rest_pixels == 0)) && pos && gw && workspace && tiles && (n_rest_pixels == 0 || rest_pixels) &&
                   (n_small == 0 || (rec_w && rec_off)) &&
                   (n_polar_items == 0 || (pitems && prec_w && prec_off)),
               MODE_ERR_BAD_ARG, "mode_sphere_conv_bwd_weight_win: null pointer");
  hipStream_t st = mode::as_stream(stream);
  const int NCG = mode::cdiv(d.Cig, BW_CG);
  const size_t slice = (size_t)d.G * d.MG * NCG * BW_SLOTS * 128 * BW_CG * sizeof(float);
  const float* gys = gy_t ? gy_t : gy;
  const float* xs = x_t ? x_t : x;
  int S = 0;
  if (n_small > 0) {
    S = bww_win_splits(d, n_small);
    if (split) {
      const size_t lds = (size_t)BS_LDS_DWORDS * sizeof(float);
      if (amax_g) {
        rc = mode::allow_lds(sphere_bww_split_kernel<true>, lds, "mode_sphere_conv_bwd_weight_win_split");
        if (rc != MODE_OK) return rc;
        hipLaunchKernelGGL(sphere_bww_split_kernel<true>, dim3(S, NCG, d.G * d.MG), dim3(NTHREADS), lds, st, gys, xs, workspace, d,
                           reinterpret_cast<const int4*>(tiles) + n_wrap + n_mid, reinterpret_cast<const float4*>(rec_w), rec_off, n_small, S,
                           amax_g, amax_x);
      } else {
        rc = mode::allow_lds(sphere_bww_split_kernel<false>, lds, "mode_sphere_conv_bwd_weight_win_split");
        if (rc != MODE_OK) return rc;
        hipLaunchKernelGGL(sphere_bww_split_kernel<false>, dim3(S, NCG, d.G * d.MG), dim3(NTHREADS), lds, st, gys, xs, workspace, d,
                           reinterpret_cast<const int4*>(tiles) + n_wrap + n_mid, reinterpret_cast<const float4*>(rec_w), rec_off, n_small, S,
                           amax_g, amax_x);
      }
    } else {
      const size_t lds = (size_t)BW_LDS_FLOATS * sizeof(float);
      rc = mode::allow_lds(sphere_bww_win_kernel, lds, "mode_sphere_conv_bwd_weight_win");
      if (rc != MODE_OK) return rc;
      hipLaunchKernelGGL(sphere_bww_win_kernel, dim3(S, NCG, d.G * d.MG), dim3(NTHREADS), lds, st, gys, xs, workspace, d,
                         reinterpret_cast<const int4*>(tiles) + n_wrap + n_mid, reinterpret_cast<const float4*>(rec_w), rec_off, n_small, S);
    }
    rc = mode::check_launch("mode_sphere_conv_bwd_weight_win");
    if (rc != MODE_OK) return rc;
  }
  int Sp = 0;
  if (n_polar_items > 0) {
    Sp = bww_polar_splits(d, n_polar_items);
    if (split) {
      const size_t lds = (size_t)BQ_LDS_DWORDS * sizeof(float);
      rc = mode::allow_lds(sphere_bww_polar_split_kernel, lds, "mode_sphere_conv_bwd_weight_win_split");
      if (rc != MODE_OK) return rc;
      hipLaunchKernelGGL(sphere_bww_polar_split_kernel, dim3(Sp, NCG, d.G * d.MG), dim3(NTHREADS), lds, st, gys, xs, workspace, d, pitems,
                         reinterpret_cast<const float4*>(prec_w), prec_off, n_polar_items, Sp, S);
    } else {
      const size_t lds = (size_t)BP_LDS_FLOATS * sizeof(float);
      rc = mode::allow_lds(sphere_bww_polar_kernel, lds, "mode_sphere_conv_bwd_weight_win");
      if (rc != MODE_OK) return rc;
      hipLaunchKernelGGL(sphere_bww_polar_kernel, dim3(Sp, NCG, d.G * d.MG), dim3(NTHREADS), lds, st, gys, xs, workspace, d, pitems,
                         reinterpret_cast<const float4*>(prec_w), prec_off, n_polar_items, Sp, S);
    }
    rc = mode::check_launch("mode_sphere_conv_bwd_weight_win(polar)");
    if (rc != MODE_OK) return rc;
  }
  if (S + Sp > 0) {
    const long long n = (long long)d.G * d.MG * NCG * KT * 128 * BW_CG;
    hipLaunchKernelGGL(reduce_gw_win, dim3(mode::cdiv(n, 256)), dim3(256), 0, st, workspace, gw, d, S + Sp, NCG);
    rc = mode::check_launch("mode_sphere_conv_bwd_weight_win(reduce)");
    if (rc != MODE_OK) return rc;
  }
  if (n_rest_pixels > 0)
    return mode::sphere_bwd_weight_general(gy, pos, x, gw, reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + slice * (S + Sp)), B,
                                           Ci, H, W, Co, Kh, Kw, 1, 1, H, W, groups, rest_pixels, n_rest_pixels, st,
                                           "mode_sphere_conv_bwd_weight_win(rest)");
  return MODE_OK;
}

// out (P, W, H) = in (P, H, W) transposed plane by plane (P = B*C).  The windowed kernels run with every access coalesced when
// the planes are stored with the shift-invariant (longitude) axis contiguous, which for the Cassini layout is the transpose.
extern "C" int mode_transpose_planes(const float* in, float* out, long long planes, int H, int W, mode_stream_t stream) {
  MODE_REQUIRE(planes >= 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG, "mode_transpose_planes: bad size");
  if (planes == 0) return MODE_OK;
  MODE_REQUIRE(in && out, MODE_ERR_BAD_ARG, "mode_transpose_planes: null pointer");
  hipStream_t st = mode::as_stream(stream);
  for (long long p0 = 0; p0 < planes; p0 += 65535) {
    const int np = (int)std::min<long long>(65535, planes - p0);
    hipLaunchKernelGGL(transpose_planes_kernel, dim3(mode::cdiv(W, 32), mode::cdiv(H, 32), np), dim3(256), 0, st,
                       in + p0 * (long long)H * W, out + p0 * (long long)H * W, H, W);
  }
  return mode::check_launch("mode_transpose_planes");
}

// =====================================================================================================================
// Input gradient on the split-bf16 matrix path: the ADJOINT of the sampling as a windowed gather.
//
//   gx[c][q] = sum_{k, o} W[o][c][k] * G_k[o][q],     G_k[o][q] = sum_{(p, wt) in L(k, q)} wt * gy[o][p]
//
// L(k, q) = the output pixels whose tap k touches input pixel q (the transpose of the sampling table; cu:293-356 scatters the same
// terms with atomics).  Away from the poles the table moves slowly, so L(k, q) has at most four entries and the sources of a 64 x 4
// block of q fall into the same compact window of gy that the forward kernel stages of x.  The host plans every tile from the actual
// table (mode_sphere_adjplan_build): tiles whose lists all have <= 4 entries inside an 81-row x 8-column window get records of
// 4 (window offset, weight) slots per (pixel, tap) and run here -- the structure of sphere_fwd_split_kernel, with the sampling
// record read from the plan instead of computed from the table: gy window of 16 channels in LDS (fp32, double-buffered), every wave
// samples its 32 pixels for the next tap into an LDS operand buffer (exact 3-way bf16 split), one output-channel tile per wave for the
// MFMAs.  All other tiles (next to the poles, and the few columns where a list has 5 or 6 entries) stay on the gather kernel of
// sphere_conv.hip, restricted to a tile list (mode_sphere_conv_bwd_data_adj_list).
namespace {

constexpr int AJ_PIX = TH * TW;  // 256 pixels per tile = 8 waves x 32 lanes

// NS = 4: every adjoint list of the tile has at most four entries; NS = 6: up to six (the columns around the equator, where the
// sampling positions of neighbouring output columns straddle a pixel boundary): slots 4 and 5 come from a second record array, read
// when the tap is sampled (one tile in sixteen: not worth prefetch registers in a kernel at 249 VGPRs).
// F16: the two-piece fp16 arithmetic of the training forward (sphere_fwd_split_kernel<false, true>): sx = the power-of-two scale of gy
// (on the record's weights when a record becomes current), unscale = 1 / (sx * the weights' scale) on the accumulators.
template <int NS, bool F16>
__device__ __forceinline__ void bwd_data_tile(const float* __restrict__ gy, const uint4* __restrict__ wps, float* __restrict__ gx,
                                              const WinDims& d /* roles swapped: Ci = gy channels */, int NCH16, const int4 t,
                                              const int4* __restrict__ rec_off, const float4* __restrict__ rec_w,
                                              const int2* __restrict__ rec_off2, const float2* __restrict__ rec_w2, float* smem,
                                              float sx, float unscale) {
  constexpr int WRP = WR_SMALL, CP = SP_CP;
  uint4* opbuf = reinterpret_cast<uint4*>(smem + SP_WIN_FLOATS);  // [3][8 pixel groups][3 pieces][64 lanes]
  const int h0 = t.x, w0 = t.y, rbase = t.z, cbase = t.w & 0xffff;
  const int b = blockIdx.y;
  const int g = blockIdx.z / d.MG, mg = blockIdx.z % d.MG;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5;
  const long long HW = (long long)d.H * d.W;

  // records of the pixel this lane SAMPLES (column wave % 4, row block wave / 4, row lane & 31), one (int4, float4) per tap
  // (uniform bases + one 32-bit lane index: four per-lane 64-bit pointers were 8 registers the six-source tiles do not have)
  const long long rtile = (long long)blockIdx.x * KT * AJ_PIX;
  const int4* rop = rec_off + rtile;
  const float4* rwp = rec_w + rtile;
  const int2* rop2 = rec_off2 + rtile;
  const float2* rwp2 = rec_w2 + rtile;
  const unsigned rlane = wave * 32 + (lane & 31);
  // records: taps 0 and 1 for the two prologue samples, tap 2 for the first tap of the loop; inside the loop the record of the tap
  // after the next sampled one is requested at the top of a tap (L2-resident, shared by all samples and layers of the geometry)
  auto record = [&](int kk, int4& o4, float4& w4, int2& o2, float2& w2) {
    // (the lane index goes through an opaque asm: otherwise the 9 x 4 per-tap addresses are hoisted out of the tap loop as 64-bit
    // per-lane pointers -- 72 registers, spilled and reloaded with a full vmcnt(0) in front of every record load)
    unsigned l = rlane;
    asm volatile("" : "+v"(l));
    o4 = rop[kk * AJ_PIX + l];
    w4 = rwp[kk * AJ_PIX + l];
    o2 = make_int2(0, 0);
    w2 = make_float2(0.f, 0.f);
    if (NS == 6) {
      o2 = rop2[kk * AJ_PIX + l];
      w2 = rwp2[kk * AJ_PIX + l];
    }
  };
  // (F16) the operand's power-of-two scale rides on the record's weights: sx * (sum of w_i g_i) bit for bit
  auto scaled = [&](float4& w4, float2& w2) {
    if (F16) {
      w4 = make_float4(w4.x * sx, w4.y * sx, w4.z * sx, w4.w * sx);
      if (NS == 6) w2 = make_float2(w2.x * sx, w2.y * sx);
    }
  };
  int4 o_p0, o_p1, o_use, o_next;
  float4 w_p0, w_p1, w_use, w_next;
  int2 o2_p0, o2_p1, o2_use, o2_next;
  float2 w2_p0, w2_p1, w2_use, w2_next;
  record(0, o_p0, w_p0, o2_p0, w2_p0);
  record(1, o_p1, w_p1, o2_p1, w2_p1);
  record(2, o_use, w_use, o2_use, w2_use);
  scaled(w_p0, w2_p0);
  scaled(w_p1, w2_p1);
  scaled(w_use, w2_use);

  // every window word that may be read is finite: the staging stores write all rows and columns of both windows; what they never write
  // is the pad word behind each channel, the slack behind the two buffers, and -- during the first chunk -- the head of the second
  // window, into which the last channel of the first runs over (all read only under zero weights)
  if (tid < 2 * SP_CCH) smem[(tid / SP_CCH) * SP_WIN + (tid % SP_CCH) * CP + WC * WRP] = 0.f;
  if (tid < WRP + 8) smem[SP_WIN + tid] = 0.f;
  for (int i = 2 * SP_WIN + tid; i < SP_WIN_FLOATS; i += NTHREADS) smem[i] = 0.f;
  // (no barrier: see sphere_fwd_split_kernel -- the one behind the first window's stores is enough)

  // window staging exactly as in sphere_fwd_split_kernel
  const int scol = d.sh == 1 ? tid / SROWS : tid & (WC - 1), srow = d.sh == 1 ? tid % SROWS : tid / WC;
  const int gcol = cbase + scol;
  const bool col_ok = gcol < d.W;
  const float* xg = gy + ((long long)b * d.Ci + (long long)g * d.Cig) * HW;
  unsigned rowoff[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int r = rb * SROWS + srow;
    rowoff[rb] = 4u * (unsigned)(((rbase + (r < WRP ? r : 0)) % d.H) * d.sh + (col_ok ? gcol * d.sw : 0));
  }
  const long long plane_bytes = 4 * HW;
  float lv[8][4];
  auto issue = [&](int ch, int ph, int set) {
    const char* xc = reinterpret_cast<const char*>(xg) + ((long long)ch * SP_CCH + ph * 2) * plane_bytes;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) lv[set][cc * 2 + rb] = *reinterpret_cast<const float*>(xc + cc * plane_bytes + rowoff[rb]);
  };
  auto commit = [&](int ch, int ph, int set, float* buf) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int c = ph * 2 + cc;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const int r = rb * SROWS + srow;
        float* dst = r < WRP ? buf + c * CP + scol * WRP + r : smem + 2 * SP_WIN + (tid & 7);
        *dst = col_ok ? lv[set][cc * 2 + rb] : 0.f;
      }
    }
  };
  // B fragment of this lane's pixel for one tap: G_k[o][q] for 8 channels o, <= 4 sources each, split, stored as this wave's group
  auto sample = [&](const float* win, const int4 o4, const float4 tw, const int2 o2, const float2 t2, uint4* op) {
    const float* p = win + half * 8 * CP;
    float v[8], r_[8][NS];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float* q = p + c * CP;
      r_[c][0] = q[o4.x];
      r_[c][1] = q[o4.y];
      r_[c][2] = q[o4.z];
      r_[c][3] = q[o4.w];
      if (NS == 6) {
        r_[c][4] = q[o2.x];
        r_[c][5] = q[o2.y];
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // (all window words requested before the first one is used)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      v[c] = __builtin_fmaf(tw.w, r_[c][3], __builtin_fmaf(tw.z, r_[c][2], __builtin_fmaf(tw.y, r_[c][1], tw.x * r_[c][0])));
      if (NS == 6) v[c] = __builtin_fmaf(t2.y, r_[c][NS - 1], __builtin_fmaf(t2.x, r_[c][NS - 2], v[c]));
      asm("" : "+v"(v[c]));  // (keeps the chains of two channels from being SLP-packed pairwise, see sphere_fwd_split_kernel)
    }
    uint32_t q1[4], q2[4], q3[4];
    uint4* dst = op + (wave * 3) * 64 + lane;
    if constexpr (F16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) sp_split2_f16(v[2 * j], v[2 * j + 1], q1[j], q2[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) sp_split2(v[2 * j], v[2 * j + 1], q1[j], q2[j], q3[j]);
      dst[128] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
    }
    dst[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    dst[64] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
  };

  f32x16 acc[4];
#pragma unroll
  for (int gi = 0; gi < 4; ++gi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[gi][r] = 0.f;
  const int m = wave % TW, gset = (wave / TW) * 4;
  const uint4* wpa = wps + ((long long)(g * d.MG + mg) * NCH16) * KT * MTW * 192 + m * 192;
  const int nsteps = NCH16 * KT;
  constexpr int NPC = F16 ? 2 : 3;  // pieces per value
  uint4 acur[3], anxt[3];  // weight fragments of this tap and of the next one
#pragma unroll
  for (int p = 0; p < NPC; ++p) acur[p] = wpa[(unsigned)(p * 64 + lane)];

#pragma unroll
  for (int ph = 0; ph < 8; ++ph) issue(0, ph, ph);
#pragma unroll
  for (int ph = 0; ph < 8; ++ph) commit(0, ph, ph, smem);
  __syncthreads();
  // ---- the tap loop: the slot schedule of sphere_fwd_split_kernel (three operand buffers, sampling two taps ahead, one fragment set
  // refilled in place, window words in two half batches of 4 channels); here a sample has NS sources at arbitrary window offsets
  sample(smem, o_p0, w_p0, o2_p0, w2_p0, opbuf);
  sample(smem, o_p1, w_p1, o2_p1, w2_p1, opbuf + SP_OP);
  sp_lds_barrier();
  uint4 b0[4], b0n[4], b1[4], b2[4];
  const uint4* fragbase = opbuf + gset * 192 + lane;
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) b0[gi] = fragbase[(gi * 3 + 0) * 64];
  constexpr int LAG = NS == 6 ? 1 : 2;  // (six-source tiles hold 8 more window words and 8 more record words: one tap less of staging in flight)
  float raw[4][NS], v[8], ra[4], rb[4];
  uint32_t q1[4], q2[4], q3[4];
  auto words = [&](const float* win, const int4 o4, const int2 o2, int hb) {
    const float* p = win + half * 8 * CP;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float* q = p + (hb * 4 + c) * CP;
      raw[c][0] = q[o4.x];
      raw[c][1] = q[o4.y];
      raw[c][2] = q[o4.z];
      raw[c][3] = q[o4.w];
      if (NS == 6) {
        raw[c][4] = q[o2.x];
        raw[c][5] = q[o2.y];
      }
    }
  };
  words(smem, o_use, o2_use, 0);
#define MODE_SB __builtin_amdgcn_sched_barrier(0);
  if constexpr (F16) {
    // ---- 12 slots, as in sphere_fwd_split_kernel<false, true>: hi x hi, lo x hi, hi x lo; the second half batch of window words in
    // registers of its own, requested at the top of the tap.  (The empty asm behind each MFMA pins it to its slot: without a consumer in
    // the slot the compiler sinks the first eight below all their sched_barriers, to where their operands' registers are assembled.)
#define MODE_MFH(PA, B, gi) acc[gi] = sp_mfma_f16(acur[PA], B[gi], acc[gi]); asm volatile("" :: "v"(acc[gi]));
    float raw2[4][NS];
    // Three record sets, the record of sampled tap s in set s % 3, requested FOUR taps ahead (at the top of tap s - 6 ... i.e. under tap
    // k the record of tap k + 4 is requested and the one of tap k + 3 becomes current in slot 8): requested one tap ahead as in the
    // three-piece loop it would be the newest request when slot 8 needs it -- a vmcnt(0) that also drains the window staging loads
    // of the previous tap, which have two taps to arrive.
    int4 ro[3];
    float4 rwt[3];
    int2 ro2[3];
    float2 rwt2[3];
    ro[2] = o_use; rwt[2] = w_use; ro2[2] = o2_use; rwt2[2] = w2_use;  // (already scaled)
    record(3, ro[0], rwt[0], ro2[0], rwt2[0]);
    for (int ch = 0; ch < NCH16; ++ch) {
      float* cur = smem + (ch & 1) * SP_WIN;
      float* nxt = smem + ((ch + 1) & 1) * SP_WIN;
      const int chn = ch + 1 < NCH16 ? ch + 1 : ch;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const int step = ch * KT + k;
        const int nstep = step + 1 < nsteps ? step + 1 : nsteps - 1;
        const float* wsrc = k + 2 < KT ? cur : nxt;
        const float* wsrcn = k + 3 < KT ? cur : nxt;
        const uint4* fcur = fragbase + (k % 3) * SP_OP;
        const uint4* fnxt = fragbase + ((k + 1) % 3) * SP_OP;
        uint4* opw = opbuf + ((k + 2) % 3) * SP_OP + (wave * 3) * 64 + lane;
        const int su = (k + 2) % 3, sn = (k + 3) % 3, sl = (k + 4) % 3;  // sets: current, next, the one being requested
        auto combine = [&](int c, float (&rr)[4][NS]) {
          const float4 tw = rwt[su];
          float t = __builtin_fmaf(tw.w, rr[c & 3][3], __builtin_fmaf(tw.z, rr[c & 3][2], __builtin_fmaf(tw.y, rr[c & 3][1], tw.x * rr[c & 3][0])));
          if (NS == 6) t = __builtin_fmaf(rwt2[su].y, rr[c & 3][NS - 1], __builtin_fmaf(rwt2[su].x, rr[c & 3][NS - 2], t));
          v[c] = t;
          asm("" : "+v"(v[c]));
        };
        auto hsplit = [&](int j) { sp_split2_f16(v[2 * j], v[2 * j + 1], q1[j], q2[j]); };
#pragma unroll
        for (int p = 0; p < 2; ++p) anxt[p] = (wpa + (long long)nstep * MTW * 192)[(unsigned)(p * 64 + lane)];
        record((k + 4) % KT, ro[sl], rwt[sl], ro2[sl], rwt2[sl]);
        {
          const float* p = wsrc + half * 8 * CP;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float* q = p + (4 + c) * CP;
            raw2[c][0] = q[ro[su].x];
            raw2[c][1] = q[ro[su].y];
            raw2[c][2] = q[ro[su].z];
            raw2[c][3] = q[ro[su].w];
            if (NS == 6) {
              raw2[c][4] = q[ro2[su].x];
              raw2[c][5] = q[ro2[su].y];
            }
          }
        }
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) b1[gi] = fcur[(gi * 3 + 1) * 64];
        MODE_SB
        MODE_MFH(0, b0, 0) combine(0, raw); MODE_SB
        MODE_MFH(0, b0, 1) combine(1, raw); MODE_SB
        MODE_MFH(0, b0, 2) combine(2, raw); MODE_SB
        MODE_MFH(0, b0, 3) combine(3, raw); MODE_SB
        MODE_MFH(1, b0, 0) hsplit(0); MODE_SB
        MODE_MFH(1, b0, 1) hsplit(1); combine(4, raw2); MODE_SB
        MODE_MFH(1, b0, 2) combine(5, raw2); combine(6, raw2); MODE_SB
        MODE_MFH(1, b0, 3) combine(7, raw2); MODE_SB
        // the record of the tap sampled under the NEXT tap becomes current; its first half batch of window words is requested here
        scaled(rwt[sn], rwt2[sn]);
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) b0n[gi] = fnxt[(gi * 3 + 0) * 64];
        words(wsrcn, ro[sn], ro2[sn], 0);
        MODE_SB
        MODE_MFH(0, b1, 0) hsplit(2); MODE_SB
        MODE_MFH(0, b1, 1) hsplit(3); MODE_SB
        opw[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
        opw[64] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
        if (k < 4) {
          issue(chn, 2 * k, 2 * k);
          issue(chn, 2 * k + 1, 2 * k + 1);
        }
        MODE_SB
        MODE_MFH(0, b1, 2) MODE_SB
        if (k >= LAG && k < 4 + LAG) commit(chn, 2 * (k - LAG), 2 * (k - LAG), nxt);
        MODE_SB
        MODE_MFH(0, b1, 3) MODE_SB
        if (k >= LAG && k < 4 + LAG) commit(chn, 2 * (k - LAG) + 1, 2 * (k - LAG) + 1, nxt);
        MODE_SB
#pragma unroll
        for (int p = 0; p < 2; ++p) acur[p] = anxt[p];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) b0[gi] = b0n[gi];
        sp_lds_barrier();
      }
    }
#undef MODE_MFH
  } else {
#define MODE_MF(PA, B, gi) acc[gi] = sp_mfma(acur[PA], B[gi], acc[gi]);

  for (int ch = 0; ch < NCH16; ++ch) {
    float* cur = smem + (ch & 1) * SP_WIN;
    float* nxt = smem + ((ch + 1) & 1) * SP_WIN;
    const int chn = ch + 1 < NCH16 ? ch + 1 : ch;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const int step = ch * KT + k;
      const int nstep = step + 1 < nsteps ? step + 1 : nsteps - 1;
      const float* wsrc = k + 2 < KT ? cur : nxt;           // window of the tap sampled under this one (tap (k + 2) % 9)
      const float* wsrcn = k + 3 < KT ? cur : nxt;          // ... and of the one sampled under the next tap
      const uint4* fcur = fragbase + (k % 3) * SP_OP;
      const uint4* fnxt = fragbase + ((k + 1) % 3) * SP_OP;
      uint4* opw = opbuf + ((k + 2) % 3) * SP_OP + (wave * 3) * 64 + lane;
      auto combine = [&](int c) {
        float t = __builtin_fmaf(w_use.w, raw[c & 3][3], __builtin_fmaf(w_use.z, raw[c & 3][2], __builtin_fmaf(w_use.y, raw[c & 3][1], w_use.x * raw[c & 3][0])));
        if (NS == 6) t = __builtin_fmaf(w2_use.y, raw[c & 3][NS - 1], __builtin_fmaf(w2_use.x, raw[c & 3][NS - 2], t));
        v[c] = t;
        asm("" : "+v"(v[c]));  // (scalar chains: see sp_split2)
      };
      auto split_a = [&](int j) {
        q1[j] = sp_pack2(v[2 * j], v[2 * j + 1]);
        ra[j] = v[2 * j] - __builtin_bit_cast(float, q1[j] << 16);
        rb[j] = v[2 * j + 1] - __builtin_bit_cast(float, q1[j] & 0xffff0000u);
        asm("" : "+v"(ra[j]), "+v"(rb[j]));
      };
      auto split_b = [&](int j) {
        q2[j] = sp_pack2(ra[j], rb[j]);
        ra[j] = ra[j] - __builtin_bit_cast(float, q2[j] << 16);
        rb[j] = rb[j] - __builtin_bit_cast(float, q2[j] & 0xffff0000u);
        asm("" : "+v"(ra[j]), "+v"(rb[j]));
      };
      auto split_c = [&](int j) { q3[j] = sp_pack2(ra[j], rb[j]); };

#pragma unroll
      for (int p = 0; p < 3; ++p) anxt[p] = (wpa + (long long)nstep * MTW * 192)[(unsigned)(p * 64 + lane)];
      record((k + 3) % KT, o_next, w_next, o2_next, w2_next);
      MODE_SB
      MODE_MF(2, b0, 0) combine(0); MODE_SB
      MODE_MF(2, b0, 1) combine(1); MODE_SB
      MODE_MF(2, b0, 2) combine(2); MODE_SB
      MODE_MF(2, b0, 3) combine(3); MODE_SB
      words(wsrc, o_use, o2_use, 1);
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) b1[gi] = fcur[(gi * 3 + 1) * 64];
      MODE_SB
      MODE_MF(1, b0, 0) split_a(0); MODE_SB
      MODE_MF(1, b0, 1) split_b(0); MODE_SB
      MODE_MF(1, b0, 2) split_c(0); split_a(1); MODE_SB
      MODE_MF(1, b0, 3) split_b(1); MODE_SB
      MODE_MF(0, b0, 0) split_c(1); combine(4); MODE_SB
      MODE_MF(0, b0, 1) combine(5); MODE_SB
      MODE_MF(0, b0, 2) combine(6); MODE_SB
      MODE_MF(0, b0, 3) combine(7); MODE_SB
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) b2[gi] = fcur[(gi * 3 + 2) * 64];
      MODE_SB
      MODE_MF(1, b1, 0) split_a(2); MODE_SB
      MODE_MF(1, b1, 1) split_b(2); MODE_SB
      MODE_MF(1, b1, 2) split_c(2); split_a(3); MODE_SB
      MODE_MF(1, b1, 3) split_b(3); MODE_SB
      MODE_MF(0, b1, 0) split_c(3); MODE_SB
      opw[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
      opw[64] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
      opw[128] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
      MODE_SB
      MODE_MF(0, b1, 1) MODE_SB
      if (k < 4) {
        issue(chn, 2 * k, 2 * k);
        issue(chn, 2 * k + 1, 2 * k + 1);
      }
      MODE_SB
      MODE_MF(0, b1, 2) MODE_SB
      MODE_MF(0, b1, 3) MODE_SB
      // the record of the tap sampled under the NEXT tap becomes current; its first half batch of window words is requested here
      o_use = o_next;
      w_use = w_next;
      o2_use = o2_next;
      w2_use = w2_next;
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) b0n[gi] = fnxt[(gi * 3 + 0) * 64];
      words(wsrcn, o_use, o2_use, 0);
      MODE_SB
      MODE_MF(0, b2, 0) MODE_SB
      if (k >= LAG && k < 4 + LAG) commit(chn, 2 * (k - LAG), 2 * (k - LAG), nxt);  // LAG taps after their loads were issued
      MODE_SB
      MODE_MF(0, b2, 1) MODE_SB
      if (k >= LAG && k < 4 + LAG) commit(chn, 2 * (k - LAG) + 1, 2 * (k - LAG) + 1, nxt);
      MODE_SB
      MODE_MF(0, b2, 2) MODE_SB
      MODE_MF(0, b2, 3) MODE_SB
#pragma unroll
      for (int p = 0; p < 3; ++p) acur[p] = anxt[p];
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) b0[gi] = b0n[gi];
      sp_lds_barrier();
    }
  }
#undef MODE_MF
  }
#undef MODE_SB

  // D[i = c][j = pixel of group gset + gi]: written (this kernel owns every element of its tiles)
  const int hh = h0 + (wave / TW) * 32 + (lane & 31);
  const int cmax = d.Cog - mg * 128;
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) {
    const int ww = w0 + gi;
    if (hh < d.H && ww < d.W) {
      float* yb = gx + ((long long)b * d.Co + (long long)g * d.Cog + (long long)mg * 128) * HW + (long long)hh * d.sh + (long long)ww * d.sw;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < cmax) yb[(long long)co * HW] = F16 ? acc[gi][r] * unscale : acc[gi][r];
      }
    }
  }
}

template <bool F16>
__global__ __launch_bounds__(NTHREADS) void sphere_bwd_data_split_kernel(const float* __restrict__ gy, const uint4* __restrict__ wps,
                                                                         float* __restrict__ gx, WinDims d, int NCH16,
                                                                         const int4* __restrict__ tiles, const int4* __restrict__ rec_off,
                                                                         const float4* __restrict__ rec_w, const int2* __restrict__ rec_off2,
                                                                         const float2* __restrict__ rec_w2, const float* __restrict__ amax_g,
                                                                         const float* __restrict__ amax_w) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float sx = 1.f, unscale = 1.f;
  if (F16) {
    sx = sp_f16_scale_of(mode::absmax_load(amax_g));
    unscale = (1.f / sx) * (1.f / sp_f16_scale_of(mode::absmax_load(amax_w)));
  }
  const int4 t = tiles[blockIdx.x];
  if ((t.w >> 16) == 0)
    bwd_data_tile<4, F16>(gy, wps, gx, d, NCH16, t, rec_off, rec_w, rec_off2, rec_w2, smem, sx, unscale);
  else
    bwd_data_tile<6, F16>(gy, wps, gx, d, NCH16, t, rec_off, rec_w, rec_off2, rec_w2, smem, sx, unscale);
}

// wps[(((((g*MG + mg)*NCH16 + ch)*KT + tap)*MTW + m)*3 + piece)*64 + lane] = 8 bf16: piece of W[o = g*Cog' + ch*16 + 8*(lane>>5) + j]
// [c = mg*128 + m*32 + (lane&31)][tap] -- the transposed weight: rows of this GEMM are the INPUT channels c of the layer, its reduction
// runs over the output channels o.  `d` carries the swapped roles (d.Cog = input channels per group, d.Cig = output channels per group).
template <bool F16>
__global__ void pack_w_win_split_t(const float* __restrict__ w, uint4* __restrict__ wps, WinDims d, int NCH16, const float* __restrict__ amax_w) {
  const float sw = F16 ? sp_f16_scale_of(mode::absmax_load(amax_w)) : 1.f;
  const long long total = (long long)d.G * d.MG * NCH16 * KT * MTW * 64;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    long long r = idx >> 6;
    const int m = (int)(r % MTW);
    r /= MTW;
    const int tap = (int)(r % KT);
    r /= KT;
    const int ch = (int)(r % NCH16);
    r /= NCH16;
    const int mg = (int)(r % d.MG);
    const int g = (int)(r / d.MG);
    const int c = mg * 128 + m * 32 + (lane & 31);  // row of the GEMM = input channel of the layer (within the group)
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int o = ch * SP_CCH + 8 * (lane >> 5) + j;  // reduction index = output channel of the layer (within the group)
      v[j] = (c < d.Cog && o < d.Cig) ? w[((long long)(g * d.Cig + o) * d.Cog + c) * KT + tap] : 0.f;
      if (F16) v[j] *= sw;
    }
    uint32_t q1[4], q2[4], q3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (F16) {
        sp_split2_f16(v[2 * j], v[2 * j + 1], q1[j], q2[j]);
        q3[j] = 0u;
      } else {
        sp_split2(v[2 * j], v[2 * j + 1], q1[j], q2[j], q3[j]);
      }
    }
    uint4* dst = wps + (idx - lane) * 3 + lane;
    dst[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    dst[64] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
    dst[128] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
  }
}

}  // namespace

// Host-side plan of the adjoint windows (stride 1, output grid = input grid).  For every 64 x 4 tile of INPUT pixels q (same tiling as
// mode_sphere_plan_build) it collects L(k, q) for all nine taps and all pixels of the tile.  A tile is "good" when every list has at
// most 4 entries and all their source pixels lie in one window of WR_SMALL rows x WC columns:
//   good_tiles[4 i ..] = (h0, w0, rbase, cbase | six << 16), six = 1 when some list of the tile has 5 or 6 entries (slots 4, 5 in
//   rec_off2 / rec_w2 [((i * 9 + tap) * 256 + pixel) * 2 + slot - 4]; lists longer than 6 make a tile bad);
//   rec_off / rec_w [((i * 9 + tap) * 256 + pixel) * 4 + slot], pixel =
//   ((rowblock * 4 + column) * 32 + row) -- the lane order of the kernel; offset = (source column - cbase) * WR_SMALL + (source row -
//   rbase) mod H, unused slots (0, 0.0f); slots in ascending source-pixel order (the gather kernel's summation order).
//   bad_tiles[2 j ..] = (h0, w0) of the others.  counts = (good, bad).
extern "C" int mode_sphere_adjplan_build(const float* pos_host, int H, int W, int Kh, int Kw, int32_t* good_tiles, int32_t* bad_tiles,
                                         int32_t* counts, int32_t* rec_off_host, float* rec_w_host, int32_t* rec_off2_host,
                                         float* rec_w2_host) {
  MODE_REQUIRE(pos_host && good_tiles && bad_tiles && counts && rec_off_host && rec_w_host && rec_off2_host && rec_w2_host,
               MODE_ERR_BAD_ARG, "mode_sphere_adjplan_build: null pointer");
  MODE_REQUIRE(H > 0 && W > 0 && Kh * Kw == KT, MODE_ERR_BAD_ARG, "mode_sphere_adjplan_build: needs a positive size and %d taps", KT);
  const long long HW = (long long)H * W;
  MODE_REQUIRE((long long)KT * HW * 4 < (1ll << 31), MODE_ERR_UNSUPPORTED, "mode_sphere_adjplan_build: table too large");
  // adjoint lists in CSR form, rows (tap, q), filled in ascending p (the order of mode_sphere_adjoint_build)
  std::vector<int32_t> rowptr((size_t)KT * HW + 1, 0);
  auto corners = [&](int k, long long p, int qs[4], float ws[4]) -> int {
    const float h = pos_host[(long long)(2 * k) * HW + p], w = pos_host[(long long)(2 * k + 1) * HW + p];
    if (!(h > -1.f && w > -1.f && h < (float)H && w < (float)W)) return 0;
    const float hf = floorf(h), wf = floorf(w);
    const int hl = (int)hf, wl = (int)wf, hh = hl + 1, wh = wl + 1;
    const float lh = h - hf, lw = w - wf, uh = 1.f - lh, uw = 1.f - lw;
    const float wt[4] = {uh * uw, uh * lw, lh * uw, lh * lw};
    const int hc[4] = {hl, hl, hh, hh}, wc[4] = {wl, wh, wl, wh};
    int n = 0;
    for (int i = 0; i < 4; ++i)
      if (hc[i] >= 0 && hc[i] <= H - 1 && wc[i] >= 0 && wc[i] <= W - 1 && wt[i] != 0.f) {
        qs[n] = hc[i] * W + wc[i];
        ws[n] = wt[i];
        ++n;
      }
    return n;
  };
  int qs[4];
  float ws[4];
  for (int k = 0; k < KT; ++k)
    for (long long p = 0; p < HW; ++p) {
      const int n = corners(k, p, qs, ws);
      for (int i = 0; i < n; ++i) rowptr[(size_t)k * HW + qs[i] + 1]++;
    }
  for (size_t i = 0; i < (size_t)KT * HW; ++i) rowptr[i + 1] += rowptr[i];
  std::vector<int32_t> ep(rowptr.back());
  std::vector<float> ew(rowptr.back());
  {
    std::vector<int32_t> cur(rowptr.begin(), rowptr.end() - 1);
    for (int k = 0; k < KT; ++k)
      for (long long p = 0; p < HW; ++p) {
        const int n = corners(k, p, qs, ws);
        for (int i = 0; i < n; ++i) {
          const int32_t at = cur[(size_t)k * HW + qs[i]]++;
          ep[at] = (int32_t)p;
          ew[at] = ws[i];
        }
      }
  }
  const int nth = mode::cdiv(H, TH), ntw = mode::cdiv(W, TW);
  int ngood = 0, nbad = 0;
  struct GoodTile {
    int h0, w0, rbase, cbase, six;
  };
  std::vector<GoodTile> good;
  for (int hg = 0; hg < nth; hg += kNumXCD)  // tile order as in mode_sphere_plan_build (tiles sharing rows meet on one XCD)
    for (int tw = 0; tw < ntw; ++tw)
      for (int hs = 0; hs < kNumXCD && hg + hs < nth; ++hs) {
        const int h0 = (hg + hs) * TH, w0 = tw * TW;
        bool ok = h0 + TH <= H && w0 + TW <= W;  // whole tiles only: ragged edges stay on the gather kernel
        bool six = false;                        // some list has 5 or 6 entries: the 6-slot class
        int dmin = 1 << 30, dmax = -(1 << 30), cmin = 1 << 30, cmax = -(1 << 30);
        for (int k = 0; k < KT && ok; ++k)
          for (int h = h0; h < h0 + TH && ok; ++h)
            for (int w = w0; w < w0 + TW; ++w) {
              const size_t row = (size_t)k * HW + (size_t)h * W + w;
              const int nent = rowptr[row + 1] - rowptr[row];
              if (nent > 6) {
                ok = false;
                break;
              }
              if (nent > 4) six = true;
              for (int e = rowptr[row]; e < rowptr[row + 1]; ++e) {
                const int hp = ep[e] / W, wp = ep[e] % W;
                int dr = (hp - h0) % H;
                if (dr > H / 2) dr -= H;
                if (dr <= -(H + 1) / 2) dr += H;
                dmin = std::min(dmin, dr);
                dmax = std::max(dmax, dr);
                cmin = std::min(cmin, wp);
                cmax = std::max(cmax, wp);
              }
            }
        int rbase = h0, cbase = std::min(w0, std::max(W - WC, 0));
        if (ok && dmax >= dmin) {
          if (dmax - dmin + 1 > WR_SMALL || cmax - cmin + 1 > WC || cmin >= (1 << 16)) ok = false;
          rbase = ((h0 + dmin) % H + H) % H;
          cbase = cmin;
        }
        if (!ok) {
          bad_tiles[2 * nbad] = h0;
          bad_tiles[2 * nbad + 1] = w0;
          ++nbad;
          continue;
        }
        good.push_back({h0, w0, rbase, cbase, six ? 1 : 0});
      }
  // the 6-slot tiles first: they are the slowest workgroups of the launch, and started first they end inside its last round
  std::stable_sort(good.begin(), good.end(), [](const GoodTile& a, const GoodTile& b2) { return a.six > b2.six; });
  for (const GoodTile& gt : good) {
    const int h0 = gt.h0, w0 = gt.w0, rbase = gt.rbase, cbase = gt.cbase;
    int32_t* tl = good_tiles + 4 * (size_t)ngood;
    tl[0] = h0; tl[1] = w0; tl[2] = rbase; tl[3] = cbase | (gt.six << 16);
    for (int k = 0; k < KT; ++k)
      for (int pix = 0; pix < AJ_PIX; ++pix) {
        const int wv = pix >> 5, h = h0 + (wv / TW) * 32 + (pix & 31), w = w0 + (wv % TW);
        const size_t row = (size_t)k * HW + (size_t)h * W + w;
        const size_t o = (((size_t)ngood * KT + k) * AJ_PIX + pix) * 4, o2 = o / 2;
        for (int s = 0; s < 4; ++s) {
          rec_off_host[o + s] = 0;
          rec_w_host[o + s] = 0.f;
        }
        rec_off2_host[o2] = rec_off2_host[o2 + 1] = 0;
        rec_w2_host[o2] = rec_w2_host[o2 + 1] = 0.f;
        int s = 0;
        for (int e = rowptr[row]; e < rowptr[row + 1]; ++e, ++s) {
          const int hp = ep[e] / W, wp = ep[e] % W;
          const int off = (wp - cbase) * WR_SMALL + ((hp - rbase) % H + H) % H;
          if (s < 4) {
            rec_off_host[o + s] = off;
            rec_w_host[o + s] = ew[e];
          } else {
            rec_off2_host[o2 + s - 4] = off;
            rec_w2_host[o2 + s - 4] = ew[e];
          }
        }
      }
    ++ngood;
  }
  counts[0] = ngood;
  counts[1] = nbad;
  return MODE_OK;
}

extern "C" size_t mode_sphere_conv_bwd_data_win_wpack_bytes(int Ci, int Co, int Kh, int Kw, int groups) {
  if (Ci <= 0 || Co <= 0 || groups <= 0 || Kh * Kw != KT || Ci % groups || Co % groups) return 0;
  const int Cig = Ci / groups, Cog = Co / groups;
  return (size_t)groups * mode::cdiv(Cig, 128) * mode::cdiv(Cog, SP_CCH) * KT * MTW * 3 * 64 * sizeof(uint4);
}

// 1 if the split kernel takes this layer: the reduction runs over the output channels of the layer in chunks of 16.
extern "C" int mode_sphere_conv_bwd_data_win_supported(int Ci, int Co, int groups) {
  return (Ci > 0 && Co > 0 && groups > 0 && Ci % groups == 0 && Co % groups == 0 && (Co / groups) % SP_CCH == 0) ? 1 : 0;
}

// gx (written, not added to) on the n_tiles good tiles of the adjoint plan; `transposed`: gy and gx are plane-transposed (B, C, W, H).
// The caller runs mode_sphere_conv_bwd_data_adj_list on the plan's bad tiles.
static int sphere_bwd_data_win_split(const char* who, const float* gy, const float* w, const float* amax_g, const float* amax_w, float* gx,
                                     float* wpack, const int32_t* tiles, int n_tiles, const int32_t* rec_off, const float* rec_w,
                                     const int32_t* rec_off2, const float* rec_w2, int B, int Ci, int H, int W, int Co, int Kh, int Kw,
                                     int groups, int transposed, mode_stream_t stream) {
  WinDims d;
  int rc = make_win_dims(d, B, Co, H, W, Ci, Kh, Kw, groups, who);  // roles swapped: this GEMM reduces over the layer's output channels
  if (rc != MODE_OK) return rc;
  MODE_REQUIRE(mode_sphere_conv_bwd_data_win_supported(Ci, Co, groups) == 1, MODE_ERR_UNSUPPORTED,
               "%s: the output channels per group (%d) must be a multiple of %d", who, Co / std::max(groups, 1), SP_CCH);
  MODE_REQUIRE(n_tiles >= 0 && (size_t)n_tiles <= mode_sphere_plan_max_tiles(H, W), MODE_ERR_BAD_ARG, "%s: bad tile count %d", who, n_tiles);
  if (transposed) {
    d.sh = 1;
    d.sw = H;
  }
  if (B == 0 || n_tiles == 0) return MODE_OK;
  MODE_REQUIRE(gy && w && gx && wpack && tiles && rec_off && rec_w && rec_off2 && rec_w2, MODE_ERR_BAD_ARG, "%s: null pointer", who);
  MODE_REQUIRE(B <= 65535 && d.G * d.MG <= 65535, MODE_ERR_UNSUPPORTED, "%s: grid limit", who);
  hipStream_t st = mode::as_stream(stream);
  const int NCH16 = d.Cig / SP_CCH;
  uint4* wps = reinterpret_cast<uint4*>(wpack);
  const long long nsplit = (long long)d.G * d.MG * NCH16 * KT * MTW * 64;
  if (amax_g) {
    if (mode::pack_needed())
      hipLaunchKernelGGL(pack_w_win_split_t<true>, dim3(mode::cdiv(nsplit, 256)), dim3(256), 0, st, w, wps, d, NCH16, amax_w);
    rc = mode::allow_lds(sphere_bwd_data_split_kernel<true>, SP3_LDS_BYTES, who);
    if (rc != MODE_OK) return rc;
    hipLaunchKernelGGL(sphere_bwd_data_split_kernel<true>, dim3(n_tiles, B, d.G * d.MG), dim3(NTHREADS), SP3_LDS_BYTES, st, gy, wps, gx, d, NCH16,
                       reinterpret_cast<const int4*>(tiles), reinterpret_cast<const int4*>(rec_off), reinterpret_cast<const float4*>(rec_w),
                       reinterpret_cast<const int2*>(rec_off2), reinterpret_cast<const float2*>(rec_w2), amax_g, amax_w);
    return mode::check_launch(who);
  }
  if (mode::pack_needed())
    hipLaunchKernelGGL(pack_w_win_split_t<false>, dim3(mode::cdiv(nsplit, 256)), dim3(256), 0, st, w, wps, d, NCH16, (const float*)nullptr);
  rc = mode::allow_lds(sphere_bwd_data_split_kernel<false>, SP3_LDS_BYTES, who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(sphere_bwd_data_split_kernel<false>, dim3(n_tiles, B, d.G * d.MG), dim3(NTHREADS), SP3_LDS_BYTES, st, gy, wps, gx, d, NCH16,
                     reinterpret_cast<const int4*>(tiles), reinterpret_cast<const int4*>(rec_off), reinterpret_cast<const float4*>(rec_w),
                     reinterpret_cast<const int2*>(rec_off2), reinterpret_cast<const float2*>(rec_w2), (const float*)nullptr,
                     (const float*)nullptr);
  return mode::check_launch(who);
}

extern "C" int mode_sphere_conv_bwd_data_win_split(const float* gy, const float* w, float* gx, float* wpack, const int32_t* tiles,
                                                   int n_tiles, const int32_t* rec_off, const float* rec_w, const int32_t* rec_off2,
                                                   const float* rec_w2, int B, int Ci, int H, int W, int Co, int Kh, int Kw, int groups,
                                                   int transposed, mode_stream_t stream) {
  return sphere_bwd_data_win_split("mode_sphere_conv_bwd_data_win_split", gy, w, nullptr, nullptr, gx, wpack, tiles, n_tiles, rec_off, rec_w,
                                   rec_off2, rec_w2, B, Ci, H, W, Co, Kh, Kw, groups, transposed, stream);
}

// The same on the two-piece fp16 arithmetic (mode_sphere_conv_fwd_win_split_f16): amax_g / amax_w = the maximum buffers of gy and of w.
extern "C" int mode_sphere_conv_bwd_data_win_split_f16(const float* gy, const float* w, const float* amax_g, const float* amax_w, float* gx,
                                                       float* wpack, const int32_t* tiles, int n_tiles, const int32_t* rec_off,
                                                       const float* rec_w, const int32_t* rec_off2, const float* rec_w2, int B, int Ci, int H,
                                                       int W, int Co, int Kh, int Kw, int groups, int transposed, mode_stream_t stream) {
  MODE_REQUIRE(amax_g && amax_w, MODE_ERR_BAD_ARG, "mode_sphere_conv_bwd_data_win_split_f16: null maximum");
  return sphere_bwd_data_win_split("mode_sphere_conv_bwd_data_win_split_f16", gy, w, amax_g, amax_w, gx, wpack, tiles, n_tiles, rec_off, rec_w,
                                   rec_off2, rec_w2, B, Ci, H, W, Co, Kh, Kw, groups, transposed, stream);
}

#ifdef MODE_TAPTIME
namespace {
__global__ void taptime_copy_kernel(unsigned long long* out, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = g_taptime[i];
}
}  // namespace
extern "C" int mode_debug_taptime(unsigned long long* dev_out, int n) {
  hipLaunchKernelGGL(taptime_copy_kernel, dim3(64), dim3(256), 0, 0, dev_out, n);
  return (int)hipDeviceSynchronize();
}
#endif
