// 3x3x3 STRIDE-2 convolution (pad 1) of the 3D regulariser on the bf16 matrix pipe with fp32 operands split exactly into three bf16
// pieces -- the arithmetic and the structure of conv3d_split.hip (read that file first), re-tiled for stride 2.
//
// Reference: hourglass conv1 / conv3 (models/mode_disparity.py:17-19: Conv3d k3 s2 p1, 32 -> 64 and 64 -> 64) and, with the roles of
// input and output gradient exchanged, the input gradient of the two transposed convolutions conv5 / conv6 (:23-25).
//
// D[i = o][j = 32 consecutive output columns wo], K of one MFMA = 8 input channels x 2 taps, 14 tap pairs per 8-channel chunk, as for
// stride 1.  What changes is the LDS tile: output voxel (od, oh, wo) reads input (2 od + kd - 1, 2 oh + kh - 1, 2 wo + kw - 1), so
//   * an output tile of 1 x 2 rows x 32 columns needs 3 x 5 input rows of 65 columns: 990 positions of 8 channels x 3 pieces = 46 KB
//     per chunk, double-buffered 96 KB (a 2 x 8-row tile as for stride 1 would need 265 KB);
//   * the B fragment of tap kw is input column 2 wo + kw for wo = 0..31 -- every second column.  Rows are therefore stored
//     de-interleaved, [33 even columns | 33 odd columns], and the three kw become the contiguous runs starting at slots 0, 33 and 1:
//     conflict-free ds_read_b128 as for stride 1.
// Four waves = 2 output-channel tiles (the stride-2 layers of the network all have 64 output channels) x 2 halves of the tap pairs
// (wave w takes the pairs of parity w / 2 for BOTH output rows: two independent accumulators to alternate between -- one accumulator
// per wave would chain 84 dependent MFMAs); the two halves of a tile are added through LDS when the tile's last chunk is done.
// 84 MFMAs per wave and chunk against 4 staged positions per thread -- four times less matrix work per staged position than stride 1,
// which is what bounds this kernel (the loads of a chunk sit under tap pairs 0..3, their split and LDS stores under pairs 10..13).
// The launched form keeps ONE LDS buffer (64 KB with the reduction area) and runs two workgroups per CU: the staging arithmetic of a
// chunk does not fit under 84 MFMAs, so it is a phase of its own that overlaps with the other workgroup's MFMA phase (template
// parameter PHASED; the double-buffered single-workgroup form of conv3d_split.hip measured 0.376 ms against 0.350).
// Persistent workgroups, XCD-contiguous tile ranges, weights split and packed once per launch: conv3d_split.hip.
#include "common.h"

#include "bn_internal.h"
#include "conv3d_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NT = 256;
constexpr int TD = 1, TH = 2;                    // output rows of a tile (depth x height); 32 output columns
constexpr int ID = 2 * TD + 1, IH = 2 * TH + 1;  // haloed input planes / rows
constexpr int IWH = 33, IW = 2 * IWH;            // 65 input columns stored as [33 even | 33 odd] (the last odd slot is unused)
constexpr int ROWS = ID * IH;                    // 15
constexpr int ITEMS = ROWS * IW;                 // 990 positions per chunk
constexpr int KIT = (ITEMS + NT - 1) / NT;       // 4 positions per thread
constexpr int PIECE = KIT * NT;                  // 1 024
constexpr int BUF = 3 * PIECE;                   // uint4 per buffer
constexpr int NPAIR = 14;
constexpr int MT = 2;                            // output-channel tiles per launch
constexpr int RED_FLOATS = 2 * TH * 16 * 64;  // the odd-pair waves' accumulators: [2 m][2 rows][16][64 lanes]

struct S2Dims {
  int B, K, Co, D, H, W;  // input volume; K = reduction channels
  int Do, Ho, Wo;
  int nWt, nHt, nDt, NCHUNK, ntiles;
};

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split2(float a, float b, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  // (the subtractions of a pair stay scalar: packed into v_pk_add_f32 each costs ~9 cycles of the MATRIX pipe -- packed fp32
  // instructions do not overlap with MFMAs on gfx950, plain ones do; tools/experiments/mfma_op_cost.hip)
  p1 = pack2(a, b);
  float ra = a - __builtin_bit_cast(float, p1 << 16), rb = b - __builtin_bit_cast(float, p1 & 0xffff0000u);
  asm("" : "+v"(ra), "+v"(rb));
  p2 = pack2(ra, rb);
  float sa = ra - __builtin_bit_cast(float, p2 << 16), sb = rb - __builtin_bit_cast(float, p2 & 0xffff0000u);
  asm("" : "+v"(sa), "+v"(sb));
  p3 = pack2(sa, sb);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// wp[(((m * NCHUNK + ch) * NPAIR + pair) * 3 + piece) * 64 + lane] = 8 bf16: piece of W(o = m*32 + (lane & 31), c = ch*8 + j,
// tap = 2 * pair + (lane >> 5)), j = 0..7; zero for tap 27, o >= rows, c >= K.  W is (rows, K, 27) as stored for both uses: the
// forward layer's weight (Co, Ci, 27), and the transposed convolution's weight (Cin, Cout, 27) read as (rows = Cin, K = Cout).
__global__ void pack_w3d_s2_split(const float* __restrict__ w, uint4* __restrict__ wp, int rows, int K, int MTr, int NCHUNK, int fold,
                                  mode_bn_epilogue bn) {
  const long long total = (long long)MTr * NCHUNK * NPAIR * 64;
  if (fold && blockIdx.x == 0) {  // eval mode: the folded BatchNorm shifts behind the packed weights (scale goes into the weights)
    float* shifts = reinterpret_cast<float*>(wp + total * 3);
    for (int o = threadIdx.x; o < rows; o += blockDim.x) shifts[o] = fold_shift(bn, o);
  }
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    long long r = idx >> 6;
    const int pair = (int)(r % NPAIR);
    r /= NPAIR;
    const int ch = (int)(r % NCHUNK);
    const int m = (int)(r / NCHUNK);
    const int o = m * 32 + (lane & 31);
    const int tap = 2 * pair + (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = ch * 8 + j;
      v[j] = (o < rows && c < K && tap < 27) ? w[((long long)o * K + c) * 27 + tap] : 0.f;
      if (fold && o < rows) v[j] *= fold_scale(bn, o);
    }
    uint32_t q1[4], q2[4], q3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], q1[j], q2[j], q3[j]);
    uint4* dst = wp + (idx - lane) * 3 + lane;
    dst[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    dst[64] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
    dst[128] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
  }
}

__host__ __device__ constexpr int tap_off(int tap) {  // LDS position offset of a tap relative to the output voxel's origin
  return ((tap / 9) * IH + (tap / 3) % 3) * IW + (tap % 3 == 0 ? 0 : tap % 3 == 1 ? IWH : 1);
}

// PHASED = false: one workgroup per CU, LDS double-buffered, the next chunk staged inside the MFMA stream (conv3d_split.hip's scheme).
// PHASED = true: ONE LDS buffer (64 KB with the reduction area), two workgroups per CU: the next chunk's loads still fly under this
// chunk's MFMAs (into registers), but their split + LDS stores form a phase of their own between two barriers -- which overlaps with
// the MFMA phase of the CU's other workgroup.  With only 84 MFMAs per wave and chunk the staging arithmetic does not fit under them
// (the stride-1 kernel has 336), and one wave per SIMD has nothing else to run while it waits.
// EPI: the eval-mode epilogue (folded BatchNorm shift, residual, ReLU) as its own instantiation.  As a run-time test inside the store loop
// (`if (epi.shift) v = apply_epi(...)`: loads through a pointer that may alias y) it made the TRAINING kernel's epilogue 32 serialised
// "LDS read -> wait -> store" round trips per lane and tile.
// (Round 5 also tried the request order that pays in the transposed kernel -- weight fragments a whole chunk ahead, the staging loads of
// chunk g + 2 issued in the commit phase of chunk g, so that no wait of a matrix phase reaches past its weights: 254 registers with 4
// spilled, and a same-box A/B of 0.392-0.398 against 0.401-0.408 ms at 32 -> 64 and 0.087-0.090 against 0.083-0.085 ms at 64 -> 64:
// not kept.)
template <bool PHASED, bool EPI>
__global__ __launch_bounds__(NT, PHASED ? 2 : 1) void conv3d_s2_split_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                                             float* __restrict__ y, S2Dims d, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) uint4 sm[];  // [PHASED ? 1 : 2][3][PIECE], then the reduction area
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m = wave & 1, kpar = wave >> 1;  // this wave's output-channel tile and the parity of the tap pairs it takes
  float* red = reinterpret_cast<float*>(sm + (PHASED ? 1 : 2) * BUF) + m * (TH * 16 * 64);

  const int nwx = gridDim.x / kNumXCD;
  const int xcd = blockIdx.x % kNumXCD, slot = blockIdx.x / kNumXCD;
  const int q = d.ntiles / kNumXCD, rr = d.ntiles % kNumXCD;
  const int t_begin = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
  const int t_count = xcd < rr ? q + 1 : q;
  const int mine = slot < t_count ? (t_count - slot + nwx - 1) / nwx : 0;
  const int G = mine * d.NCHUNK;

  const long long HW = (long long)d.H * d.W;
  const long long DHW = (long long)d.D * HW;
  const long long oHW = (long long)d.Ho * d.Wo, oDHW = (long long)d.Do * oHW;

  auto tile_of = [&](int k, int& b, int& d0, int& h0, int& w0) {  // output coordinates of the tile's first voxel
    int t = t_begin + slot + k * nwx;
    w0 = (t % d.nWt) * 32;
    t /= d.nWt;
    h0 = (t % d.nHt) * TH;
    t /= d.nHt;
    d0 = (t % d.nDt) * TD;
    b = t / d.nDt;
  };

  // staging: position k of this thread is (input plane, row, column) = (pdz, phy, pwx)[k] of the haloed tile, stored at slot tid + k NT
  int pdz[KIT], phy[KIT], pwx[KIT], poff[KIT];
#pragma unroll
  for (int k = 0; k < KIT; ++k) {
    const int item = min(tid + k * NT, ITEMS - 1);
    const int r = item / IW, s = item - r * IW;
    pwx[k] = s < IWH ? 2 * s : 2 * (s - IWH) + 1;  // (65 for the unused last odd slot: masked by the bounds test below)
    pdz[k] = r / IH;
    phy[k] = r - pdz[k] * IH;
    poff[k] = pdz[k] * (int)HW + phy[k] * d.W + pwx[k];
  }
  float raw[KIT][8];
  // Staging as in conv3d_split.hip since round 6: buffer loads -- the chunk's 8 channel planes one descriptor, a channel a
  // scalar offset, a position in the zero padding an out-of-range lane offset that reads as zero -- and the tiles walked incrementally
  // (sb, sd, sh, sw in OUTPUT-tile units) instead of a division of the tile index per chunk.
  int jw, jh, jd, jb, sw, sh, sd, sb, s_ch = 0;
  {
    int t = nwx;
    jw = t % d.nWt;
    t /= d.nWt;
    jh = t % d.nHt;
    t /= d.nHt;
    jd = t % d.nDt;
    jb = t / d.nDt;
    t = t_begin + slot;
    sw = t % d.nWt;
    t /= d.nWt;
    sh = t % d.nHt;
    t /= d.nHt;
    sd = t % d.nDt;
    sb = t / d.nDt;
  }
  unsigned soff[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) soff[c] = (unsigned)c * (unsigned)DHW * 4u;
  __amdgpu_buffer_rsrc_t st_rs = buf_rsrc(x, 0);
  int st_base = 0, st_d0 = 0, st_h0 = 0, st_w0 = 0;  // input coordinates of the haloed tile's first voxel
  auto stage_advance = [&](int step) {
    s_ch += step;
    const int wrap = s_ch >= d.NCHUNK ? 1 : 0;
    s_ch = wrap ? 0 : s_ch;
    sw += wrap ? jw : 0;
    int c = sw >= d.nWt ? 1 : 0;
    sw -= c ? d.nWt : 0;
    sh += (wrap ? jh : 0) + c;
    c = sh >= d.nHt ? 1 : 0;
    sh -= c ? d.nHt : 0;
    sd += (wrap ? jd : 0) + c;
    c = sd >= d.nDt ? 1 : 0;
    sd -= c ? d.nDt : 0;
    sb += (wrap ? jb : 0) + c;
  };
  auto stage_begin = [&]() {
    st_d0 = 2 * (sd * TD) - 1;
    st_h0 = 2 * (sh * TH) - 1;
    st_w0 = 2 * (sw * 32) - 1;
    st_rs = buf_rsrc(x + ((long long)sb * d.K + s_ch * 8) * DHW, (unsigned)DHW * 32u);
    st_base = st_d0 * (int)HW + st_h0 * d.W + st_w0;
  };
  auto stage_load = [&](int k) {
    const unsigned ok = (unsigned)((unsigned)(st_d0 + pdz[k]) < (unsigned)d.D) & (unsigned)((unsigned)(st_h0 + phy[k]) < (unsigned)d.H) &
                        (unsigned)((unsigned)(st_w0 + pwx[k]) < (unsigned)d.W) & (unsigned)(pwx[k] < 2 * IWH - 1);
    const unsigned off = ok ? (unsigned)(st_base + poff[k]) * 4u : kBufOOB;
#pragma unroll
    for (int c = 0; c < 8; ++c) raw[k][c] = buf_load_f32(st_rs, off, soff[c]);
  };
  uint32_t sq[3][4];
  auto stage_commit = [&](int buf, int k, int h) {
#pragma unroll
    for (int j = 2 * h; j < 2 * h + 2; ++j) split2(raw[k][2 * j], raw[k][2 * j + 1], sq[0][j], sq[1][j], sq[2][j]);
    if (h == 1) {
      uint4* dst = sm + buf * BUF + tid + k * NT;
#pragma unroll
      for (int p = 0; p < 3; ++p) dst[p * PIECE] = make_uint4(sq[p][0], sq[p][1], sq[p][2], sq[p][3]);
    }
  };

  f32x16 acc[TH];
#pragma unroll
  for (int r = 0; r < TH; ++r) acc[r] = (f32x16){0};
  unsigned out_mag = 0;  // (EPI) the largest finite magnitude this thread stored: the next eval layer's operand maximum (epi.amax)
  const int half = lane >> 5;
  const int rowpos = lane & 31;  // origin of this lane's output voxel of row 0 (row r: + 2 r IW; TD = 1: plane 0)
  const long long mstride = (long long)d.NCHUNK * NPAIR * 192;
  const uint4* wpm = wp + m * mstride;

  // this wave's seven tap pairs of a chunk: pair 2 i + kpar, i = 0..6; weight fragments 3 of them ahead (slot = i: seven pairs per
  // chunk keep the slots aligned from chunk to chunk; four of them are live at any time)
  uint4 aring[7][3];
  auto load_a = [&](int slot4, int ch, int i) {
    const uint4* wq = wpm + ((long long)ch * NPAIR + 2 * i + kpar) * 192 + lane;
#pragma unroll
    for (int p = 0; p < 3; ++p) aring[slot4][p] = wq[p * 64];
  };

  if (G > 0) {
    stage_begin();
#pragma unroll
    for (int k = 0; k < KIT; ++k) stage_load(k);
#pragma unroll
    for (int k = 0; k < KIT; ++k) {
      stage_commit(0, k, 0);
      stage_commit(0, k, 1);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) load_a(i, 0, i);
  }
  __syncthreads();

  // LDS offset of the two taps of pair 2 i + kpar for this half-wave (tap 27 of the last pair reads tap 26 again: zero weights)
  int toff[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int t0e = 4 * i, t1e = 4 * i + 1;                                  // kpar = 0: pair 2 i
    const int t0o = 4 * i + 2, t1o = 4 * i + 3 < 27 ? 4 * i + 3 : 26;        // kpar = 1: pair 2 i + 1
    toff[i] = kpar ? (half ? tap_off(t1o) : tap_off(t0o)) : (half ? tap_off(t1e) : tap_off(t0e));
  }

  int ch = 0, k_tile = 0;
  for (int g = 0; g < G; ++g) {
    const uint4* src = sm + (PHASED ? 0 : (g & 1) * BUF);
    const int ch_next = ch + 1 < d.NCHUNK ? ch + 1 : 0;
    stage_advance(g + 1 < G ? 1 : 0);  // (after the last chunk the same one is staged once more: no branch in the body)
    stage_begin();
    uint4 bq[2][TH][3];
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[0][r][p] = src[p * PIECE + rowpos + 2 * r * IW + toff[0]];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      if (i + 1 < 7) {
#pragma unroll
        for (int r = 0; r < TH; ++r)
#pragma unroll
          for (int p = 0; p < 3; ++p) bq[(i + 1) & 1][r][p] = src[p * PIECE + rowpos + 2 * r * IW + toff[i + 1]];
      }
      if (i + 3 < 7)
        load_a(i + 3, ch, i + 3);
      else
        load_a(i + 3 - 7, ch_next, i + 3 - 7);
      if (i < KIT) stage_load(i);  // the next chunk: loads under this wave's first four pairs, split + stores under its last ones
      if (!PHASED && i >= 7 - KIT + 1) {  // positions 0, 1 under pair 4, then one per pair (4 positions, 3 pairs left after the loads)
        stage_commit((g + 1) & 1, i - (7 - KIT + 1) + 1, 0);
        stage_commit((g + 1) & 1, i - (7 - KIT + 1) + 1, 1);
      }
      if (!PHASED && i == 7 - KIT + 1) {
        stage_commit((g + 1) & 1, 0, 0);
        stage_commit((g + 1) & 1, 0, 1);
      }
      // smallest terms first; consecutive MFMAs alternate between the two rows' accumulators
#define MODE_S2_TERM(PA, PB) _Pragma("unroll") for (int r = 0; r < TH; ++r) acc[r] = mfma_bf16(aring[i][PA], bq[i & 1][r][PB], acc[r]);
      MODE_S2_TERM(2, 0)
      MODE_S2_TERM(0, 2)
      MODE_S2_TERM(1, 1)
      MODE_S2_TERM(1, 0)
      MODE_S2_TERM(0, 1)
      MODE_S2_TERM(0, 0)
#undef MODE_S2_TERM
#pragma unroll
      for (int q2 = 0; q2 < 6 * TH; ++q2) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ch == d.NCHUNK - 1) {  // tile finished: the odd-pair waves hand their sums over, the even-pair waves add and store D[o][wo]
      if (kpar == 1) {
#pragma unroll
        for (int r = 0; r < TH; ++r)
#pragma unroll
          for (int qq = 0; qq < 16; ++qq) red[(r * 16 + qq) * 64 + lane] = acc[r][qq];
      }
      lds_barrier();
      if (kpar == 0) {
        int b, d0, h0, w0;
        tile_of(k_tile, b, d0, h0, w0);
        const int gw = w0 + (lane & 31);
        int mo = m * 32 + 4 * half;  // opaque: the 16 clamped channel offsets derived from it are loop invariants the compiler otherwise
        if (EPI) asm volatile("" : "+v"(mo));  // keeps in registers across the chunk loop (34 spilled registers)
#pragma unroll
        for (int r = 0; r < TH; ++r) {
          const int gh = h0 + r;
          if (d0 < d.Do && gh < d.Ho && gw < d.Wo) {
            float* yb = y + (long long)b * d.Co * oDHW + d0 * oHW + (long long)gh * d.Wo + gw;
            // eval mode: folded BatchNorm shift (+ residual) (+ ReLU); the shifts and residual values of HALF a row (8 channels) are
            // requested together ahead of its stores (read next to the stores -- `shift` / `add` may alias y -- every store waited for two
            // loads; a whole row at once spilled 34 registers at the two workgroups per CU this kernel runs at)
            constexpr int HB = EPI ? 4 : 8;  // channels per batch
#pragma unroll
            for (int hq = 0; hq < 16 / HB; ++hq) {
              float other[HB];  // the odd-pair wave's sums of this half row: all 8 reads in flight before the first is used
#pragma unroll
              for (int q8 = 0; q8 < HB; ++q8) other[q8] = red[(r * 16 + HB * hq + q8) * 64 + lane];
              float shv[HB], res[HB];
              if (EPI) {
#pragma unroll
                for (int q8 = 0; q8 < HB; ++q8) {
                  const int qq = HB * hq + q8;
                  const int o = min(mo + (qq & 3) + 8 * (qq >> 2), d.Co - 1);
                  shv[q8] = epi.shift[o];
                  res[q8] = 0.f;
                }
                if (epi.add) {
#pragma unroll
                  for (int q8 = 0; q8 < HB; ++q8) {
                    const int qq = HB * hq + q8;
                    const int o = min(mo + (qq & 3) + 8 * (qq >> 2), d.Co - 1);
                    res[q8] = epi.add[(yb - y) + o * oDHW];
                  }
                }
                __builtin_amdgcn_sched_barrier(0);
              }
#pragma unroll
              for (int q8 = 0; q8 < HB; ++q8) {
                const int qq = HB * hq + q8;
                const int o = mo + (qq & 3) + 8 * (qq >> 2);
                if (o < d.Co) {
                  float v = acc[r][qq] + other[q8];
                  if (EPI) {
                    v = (v + shv[q8]) + res[q8];
                    v = epi.relu ? relu_nan(v) : v;
                    out_mag = max(out_mag, mode::absmax_mag(v));
                  }
                  yb[o * oDHW] = v;
                }
              }
              if (EPI) __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < TH; ++r) acc[r] = (f32x16){0};
      ++k_tile;
    }
    ch = ch_next;
    lds_barrier();
    if (PHASED) {  // everyone is done reading the chunk: the next one goes into the same buffer
#pragma unroll
      for (int k = 0; k < KIT; ++k) {
        stage_commit(0, k, 0);
        stage_commit(0, k, 1);
      }
      lds_barrier();
    }
  }
  if (EPI) {
    if (epi.amax) {  // (uniform)
      __syncthreads();
      mode::absmax_block_commit(out_mag, epi.amax, reinterpret_cast<unsigned*>(sm));
    }
  }
}

}  // namespace

namespace mode {

size_t conv3d_s2_split_wpack_floats(int K, int rows) { return (size_t)cdiv(rows, 32) * cdiv(K, 8) * NPAIR * 3 * 64 * 4 + 64; }

// rows = output channels of this GEMM (33..64: two 32-channel tiles), K = its reduction channels (a multiple of 8); 32-bit lane offsets
bool conv3d_s2_split_supported(int K, int rows) { return rows > 32 && rows <= 64 && K > 0 && K % 8 == 0; }

int conv3d_s2_split(const float* x, const float* w, float* y, float* wpack, int B, int K, int rows, int D, int H, int W, hipStream_t st,
                    const char* who, const mode_bn_epilogue* bn, float* amax_y) {
  MODE_REQUIRE(!amax_y || bn, MODE_ERR_BAD_ARG, "%s: the output maximum belongs to the eval epilogue", who);
  MODE_REQUIRE(conv3d_s2_split_supported(K, rows), MODE_ERR_UNSUPPORTED, "%s: %d output / %d reduction channels not supported by the stride-2 split kernel",
               who, rows, K);
  MODE_REQUIRE((long long)D * H * W < (1ll << 26), MODE_ERR_UNSUPPORTED, "%s: volume beyond the 32-bit lane offsets of the split kernel", who);  // (8 planes < 2^31 bytes)
  S2Dims d;
  d.B = B; d.K = K; d.Co = rows; d.D = D; d.H = H; d.W = W;
  d.Do = (D - 1) / 2 + 1; d.Ho = (H - 1) / 2 + 1; d.Wo = (W - 1) / 2 + 1;
  d.nWt = cdiv(d.Wo, 32); d.nHt = cdiv(d.Ho, TH); d.nDt = cdiv(d.Do, TD);
  d.NCHUNK = cdiv(K, 8);
  d.ntiles = B * d.nDt * d.nHt * d.nWt;
  const long long npack = (long long)MT * d.NCHUNK * NPAIR * 64;
  if (mode::pack_needed()) hipLaunchKernelGGL(pack_w3d_s2_split, dim3(cdiv(npack, 256)), dim3(256), 0, st, w, reinterpret_cast<uint4*>(wpack), rows, K, MT, d.NCHUNK,
                     bn ? 1 : 0, bn ? *bn : mode_bn_epilogue());
  Epi epi = make_epi(bn, wpack + npack * 3 * 4);
  if (amax_y) {
    int rc = mode::absmax_begin(amax_y, st, who);
    if (rc != MODE_OK) return rc;
    epi.amax = reinterpret_cast<unsigned*>(amax_y);
  }
  constexpr size_t LDS1 = (size_t)BUF * sizeof(uint4) + (size_t)RED_FLOATS * sizeof(float);  // 65 536 B: two workgroups per CU
  auto kern = epi.shift ? conv3d_s2_split_kernel<true, true> : conv3d_s2_split_kernel<true, false>;
  int rc = mode::allow_lds(kern, LDS1, who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(kern, dim3(2 * kNumCU), dim3(NT), LDS1, st, x, reinterpret_cast<const uint4*>(wpack), y, d, epi);
  return mode::check_launch(who);
}

}  // namespace mode
