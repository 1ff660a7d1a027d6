"""Generator of tools/experiments/tap_asm.hip: the tap loop of sphere_fwd_split_kernel as a hand-placed instruction sequence.

tap_pipeline.hip (compiler-scheduled) showed that neither the LDS traffic nor the vector arithmetic nor the barrier alone explains a
tap's 2 100-2 700 cycles.  Here every instruction of a tap is an `asm volatile` in a chosen slot (24 slots = 24 MFMAs per wave), the
s_waitcnt values are computed from the issue order, and classes of instructions can be left out:

  python tools/experiments/gen_tap_asm.py > /tmp/tap_asm.hip && hipcc --offload-arch=gfx950 -O3 -o /tmp/tap_asm /tmp/tap_asm.hip && /tmp/tap_asm

Schedule of a tap (three operand buffers, one fragment set refilled in place; sampling runs two taps ahead):
  slot s in 0..3   : read piece 0 of THIS tap's fragment s;            window words of channel s     (2 x ds_read2_b32)
  slot s in 4..7   : read piece 2 of the NEXT tap's fragment s - 4;    window words of channel s     (2 x ds_read2_b32)
  slot s in 12..15 : read piece 1 of the NEXT tap's fragment s - 12
  MFMAs            : slots 0..3 term (0,2), 4..11 terms (1,1) (0,1), 12..23 terms (2,0) (1,0) (0,0)
  arithmetic       : 4 per slot from slot 3 on: bilinear combine of a channel 3 slots after its words, then the 3-way splits
  stores           : 3 x ds_write_b128 in slot WSLOT, then s_waitcnt lgkmcnt(0) + s_barrier
"""
import sys

WRP, CP = 81, 8 * 81 + 1
OPB = 8 * 3 * 64 * 16  # bytes per operand buffer


def gen_variant(name, reads=True, frags=True, valu=True, writes=True, barrier=True, wslot=22, lead=3, vpm=4, agpr=False, layout=0):
  L = []
  emit = L.append
  emit('template <> __global__ __launch_bounds__(512) void tap_asm<%s>(float* out, int ntaps) {' % name)
  emit('''  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5;
  const unsigned WINB = (((16 * %d + 81 + 8 + 3) / 4) * 4) * 4;  // bytes of the window region
  for (int i = tid; i < (int)(WINB / 4); i += 512) smem[i] = 1.0f + 1e-3f * (i %% 97);
  u32x4* opbuf = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(smem) + WINB);
  for (int i = tid; i < 3 * 8 * 3 * 64; i += 512) opbuf[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  f32x16 acc0, acc1, acc2, acc3;
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = acc3[r] = 0.f;
  u32x4 a0 = {0x3f803f80u, 0x3c003c00u, (unsigned)lane, 0x3f803f80u}, a1 = a0, a2 = a0;
  a1[0] += 1; a2[0] += 2;
  u32x4 b00, b01, b02, b10, b11, b12, b20, b21, b22, b30, b31, b32;
  b00 = b01 = b02 = b10 = b11 = b12 = b20 = b21 = b22 = b30 = b31 = b32 = u32x4{0x3f803f80u, (unsigned)lane, 0x3c003c00u, 1u};
  const float tx = 0.27f, ty = 0.25f, tz = 0.23f, tw = 0.25f;
  unsigned ad0, ad1, ad2, ad3, ad4, ad5, ad6, ad7;  // window addresses of the 8 channels of this lane
  { const unsigned base = (half * 8 * %d + (lane & 31)) * 4;
    ad0 = base; ad1 = base + %d; ad2 = base + 2 * %d; ad3 = base + 3 * %d; ad4 = base + 4 * %d; ad5 = base + 5 * %d; ad6 = base + 6 * %d; ad7 = base + 7 * %d; }
  const unsigned fr = WINB + ((wave / 4) * 4 * 3 * 64 + lane) * 16;  // fragment reads: + (gi * 3 + p) * 1024 + buffer * OPB
  const unsigned fw = WINB + (wave * 3 * 64 + lane) * 16;            // stores: + p * 1024 + buffer * OPB
  u32x2 ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7, rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7;
  ra0 = ra1 = ra2 = ra3 = ra4 = ra5 = ra6 = ra7 = rb0 = rb1 = rb2 = rb3 = rb4 = rb5 = rb6 = rb7 = u32x2{0x3f800000u, 0x3f800000u};
  float v0 = 1.f, v1 = 1.f, v2 = 1.f, v3 = 1.f, v4 = 1.f, v5 = 1.f, v6 = 1.f, v7 = 1.f;
  unsigned t1 = 0;
  unsigned p10 = 0, p11 = 0, p12 = 0, p13 = 0, p20 = 0, p21 = 0, p22 = 0, p23 = 0, p30 = 0, p31 = 0, p32 = 0, p33 = 0;
  u32x4 q1 = {0, 0, 0, 0}, q2 = q1, q3 = q1;
  for (int t = 0; t < ntaps; t += 3) {''' % (CP, CP, CP * 4, CP * 4, CP * 4, CP * 4, CP * 4, CP * 4, CP * 4))

  class Block(object):
    """One asm volatile statement: instructions with {name} placeholders, outputs (read-write) and inputs."""

    def __init__(self):
      self.lines, self.outs, self.ins = [], [], []

    def op(self, text, outs=(), ins=()):
      for o in outs:
        if o not in self.outs:
          self.outs.append(o)
      for i in ins:
        if i not in self.ins and i not in self.outs:
          self.ins.append(i)
      self.lines.append(text)

    def flush(self):
      if not self.lines:
        return
      self.ins = [i for i in self.ins if i not in self.outs]
      names = {}
      for n, o in enumerate(self.outs + self.ins):
        names[o] = '%%%d' % n
      body = []
      for ln in self.lines:
        for o in sorted(names, key=len, reverse=True):
          ln = ln.replace('{' + o + '}', names[o])
        body.append(ln)
      emit('    asm volatile("%s" : %s : %s : "memory");' % ('\\n\\t'.join(body), ', '.join('"+%s"(%s)' % ('a' if (agpr and o.startswith('acc')) else 'v', o) for o in self.outs),
                                                            ', '.join('"v"(%s)' % i for i in self.ins)))
      self.lines, self.outs, self.ins = [], [], []

  for sub in range(3):
    cur, nxt, wr = sub % 3, (sub + 1) % 3, (sub + 2) % 3
    issued = []  # LDS ops in issue order (names)
    blk = Block()

    def lds(nm, text, outs=(), ins=()):
      issued.append(nm)
      blk.op(text, outs, ins)

    def need(nm):
      if nm not in issued:
        return
      n = len(issued) - 1 - issued.index(nm)
      if n < 15:
        blk.op('s_waitcnt lgkmcnt(%d)' % n)

    work = []
    for c in range(8):
      v = 'v%d' % c
      work.append((['sb%d' % c], 'v_mul_f32 {%s}, {tx}, {ra%d[0]}' % (v, c), [v], ['tx', 'ra%d[0]' % c]))
      work.append(([], 'v_fmac_f32 {%s}, {ty}, {ra%d[1]}' % (v, c), [v], ['ty', 'ra%d[1]' % c]))
      work.append(([], 'v_fmac_f32 {%s}, {tz}, {rb%d[0]}' % (v, c), [v], ['tz', 'rb%d[0]' % c]))
      work.append(([], 'v_fmac_f32 {%s}, {tw}, {rb%d[1]}' % (v, c), [v], ['tw', 'rb%d[1]' % c]))
      if c % 2 == 1:
        j = c // 2
        a, b = 'v%d' % (c - 1), 'v%d' % c
        for piece, q in ((1, 'q1'), (2, 'q2'), (3, 'q3')):
          qe = 'p%d%d' % (piece, j)
          work.append(([], 's_nop 0\\n\\tv_cvt_pk_bf16_f32 {%s}, {%s}, {%s}' % (qe, a, b), [qe], [a, b]))
          if piece < 3:
            work.append(([], 'v_lshlrev_b32 {t1}, 16, {%s}' % qe, ['t1'], [qe]))
            work.append(([], 'v_sub_f32 {%s}, {%s}, {t1}' % (a, a), [a], ['t1']))
            work.append(([], 'v_and_b32 {t1}, 0xffff0000, {%s}' % qe, ['t1'], [qe]))
            work.append(([], 'v_sub_f32 {%s}, {%s}, {t1}' % (b, b), [b], ['t1']))
    if not valu:
      work = []
    wi = 0
    terms = [(0, 2)] * 1 + [(1, 1), (0, 1)] + [(2, 0), (1, 0), (0, 0)]
    for slot in range(24):
      pa, pb = terms[slot // 4]
      gi = slot % 4
      if pb == 0 and frags:
        need('f0_%d' % gi)
      blk.op('v_mfma_f32_32x32x16_bf16 {acc%d}, {a%d}, {b%d%d}, {acc%d}' % (gi, pa, gi, pb, gi), ['acc%d' % gi], ['a%d' % pa, 'b%d%d' % (gi, pb)])
      if slot >= lead:
        for _ in range(vpm):
          if wi < len(work):
            deps, text, outs, ins = work[wi]
            if any(d not in issued for d in deps):
              break
            for d in deps:
              if reads:
                need(d)
            blk.op(text, outs, ins)
            wi += 1
      if layout == 0:
        f0s, f2s, f1s, wins = [0, 1, 2, 3], [4, 5, 6, 7], [12, 13, 14, 15], [0, 1, 2, 3, 4, 5, 6, 7]
      elif layout == 1:
        f0s, f2s, f1s, wins = [1, 3, 5, 7], [9, 11, 13, 15], [17, 18, 19, 20], [0, 2, 4, 6, 8, 10, 12, 14]
      elif layout == 2:
        f0s, f2s, f1s, wins = [0, 2, 4, 6], [8, 10, 12, 14], [16, 17, 18, 19], [1, 3, 5, 7, 9, 11, 13, 15]
      else:
        f0s, f2s, f1s, wins = [1, 2, 4, 5], [7, 8, 10, 11], [13, 14, 16, 17], [0, 1, 3, 4, 6, 7, 9, 10]
      if frags and slot in f0s:
        g = f0s.index(slot)
        lds('f0_%d' % g, 'ds_read_b128 {b%d0}, {fr} offset:%d' % (g, cur * OPB + (g * 3 + 0) * 1024), ['b%d0' % g], ['fr'])
      if frags and slot in f2s:
        g = f2s.index(slot)
        lds('f2_%d' % g, 'ds_read_b128 {b%d2}, {fr} offset:%d' % (g, nxt * OPB + (g * 3 + 2) * 1024), ['b%d2' % g], ['fr'])
      if frags and slot in f1s:
        g = f1s.index(slot)
        lds('f1_%d' % g, 'ds_read_b128 {b%d1}, {fr} offset:%d' % (g, nxt * OPB + (g * 3 + 1) * 1024), ['b%d1' % g], ['fr'])
      if slot in wins:
        c = wins.index(slot)
        if reads:
          lds('sa%d' % c, 'ds_read2_b32 {ra%d}, {ad%d} offset1:%d' % (c, c, WRP), ['ra%d' % c], ['ad%d' % c])
          lds('sb%d' % c, 'ds_read2_b32 {rb%d}, {ad%d} offset0:1 offset1:%d' % (c, c, WRP + 1), ['rb%d' % c], ['ad%d' % c])
        else:
          issued.append('sb%d' % c)
      if slot == wslot:
        while wi < len(work):
          deps, text, outs, ins = work[wi]
          for d in deps:
            if reads:
              need(d)
          blk.op(text, outs, ins)
          wi += 1
        blk.flush()
        if writes:
          emit('    q1 = u32x4{p10, p11, p12, p13}; q2 = u32x4{p20, p21, p22, p23}; q3 = u32x4{p30, p31, p32, p33};')
          for p, q in enumerate(('q1', 'q2', 'q3')):
            lds('w%d' % p, 'ds_write_b128 {fw}, {%s} offset:%d' % (q, wr * OPB + p * 1024), [], ['fw', q])
      blk.flush()
    blk.op('s_waitcnt lgkmcnt(0)')
    if barrier:
      blk.op('s_barrier')
    blk.flush()
  emit('  }')
  emit('''  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  s += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + (float)(q1[0] + q2[1] + q3[2] + p10 + p21 + p32);
  out[blockIdx.x * 512 + tid] = s;
}
''')
  return '\n'.join(L)


HEADER = r'''// GENERATED by tools/experiments/gen_tap_asm.py -- see there.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int V> __global__ void tap_asm(float* out, int ntaps);
'''

VARIANTS = [
    ('full schedule', dict()),
    ('no window reads', dict(reads=False)),
    ('no fragment reads', dict(frags=False)),
    ('no arithmetic', dict(valu=False)),
    ('no operand stores', dict(writes=False)),
    ('no barrier', dict(barrier=False)),
    ('MFMAs + barrier only', dict(reads=False, frags=False, valu=False, writes=False)),
    ('MFMAs + arithmetic + barrier', dict(reads=False, frags=False, writes=False)),
    ('MFMAs + all LDS traffic, no arithmetic', dict(valu=False)),
    ('full, stores in slot 18', dict(wslot=18)),
    ('full, stores in slot 20, 5 arithmetic instructions per slot', dict(wslot=20, vpm=5)),
    ('full, arithmetic 2 slots behind its words', dict(lead=2)),
    ('full, arithmetic 5 slots behind its words', dict(lead=5, vpm=5)),
    ('full, window words every other slot (layout 1), stores in slot 23', dict(layout=1, wslot=23)),
    ('full, layout 2, stores in slot 23', dict(layout=2, wslot=23)),
    ('full, layout 3, stores in slot 23', dict(layout=3, wslot=23)),
    ('layout 1, no arithmetic', dict(layout=1, wslot=23, valu=False)),
    ('full schedule, accumulators in AccVGPRs', dict(agpr=True)),
    ('MFMAs + all LDS traffic, no arithmetic, accumulators in AccVGPRs', dict(valu=False, agpr=True)),
    ('MFMAs + barrier only, accumulators in AccVGPRs', dict(reads=False, frags=False, valu=False, writes=False, agpr=True)),
]


def main():
  print(HEADER)
  for i, (nm, kw) in enumerate(VARIANTS):
    print(gen_variant(str(i), **kw))
  print(r'''
template <int V> void run(float* out, const char* what) {
  const size_t lds = (size_t)(((16 * %d + 81 + 8 + 3) / 4) * 4) * 4 + 3 * (size_t)%d;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tap_asm<V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int ntaps = 720;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  tap_asm<V><<<256, 512, lds>>>(out, 18);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  tap_asm<V><<<256, 512, lds>>>(out, ntaps);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%%-64s %%.3f us per tap = %%.0f cycles at 2.0 GHz   (%%s)\n", what, ms * 1e3 / ntaps, ms * 1e3 / ntaps * 2000.0, hipGetErrorString(hipGetLastError()));
}
int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);''' % (CP, OPB))
  for i, (nm, kw) in enumerate(VARIANTS):
    print('  run<%d>(out, "%s");' % (i, nm))
  print('  return 0;\n}')


if __name__ == '__main__':
  main()
