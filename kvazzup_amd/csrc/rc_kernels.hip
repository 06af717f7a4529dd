// kvazzup_amd/csrc/rc_kernels.hip -- "uvgx rate control v2": feedback inside the picture, decided on the device.
//
// The picture-level controller (Encoder::rate_control, statement rate_control() in oracle/hevc_enc.c) sets the picture's QP from
// access-unit sizes three pictures old.  On top of it a P picture's CTU rows are reconstructed in a few groups, one launch of
// k_inter_recon each, and between two groups k_rc_band prices the levels coded so far -- level cost of a CTU = sum over its
// non-zero levels of 3 + 2 floor(log2 |level|), one unit worth ratio / 256 bits as measured on earlier P pictures -- against the
// share of the picture's target the rows done are entitled to, and moves the QP of the next group's CTUs one step (within +-3 of
// the picture's QP).  The host never sees these QPs: they go into the per-CTU QP array the quantiser, the cu_qp_delta chain
// (k_qp_first / k_qp_chain) and deblocking read anyway.  Statement of record: rc_ctu_cost(), rc_band_decide(),
// rc_picture_start() in oracle/hevc_enc.c.
#include <hip/hip_runtime.h>
#include "hevc_core.h"
#include "enc_kernels.h"
#include "kernel_common.h"

namespace kvzx {

// before the picture's first group: the access unit of three pictures ago has been sized (bits3); if that picture was coded in groups
// its bits per unit of level cost update the ratio
__global__ void k_rc_begin(RcState *rc, uint32_t bits3, int slot3, int have3)
{
  rc->cost_sofar = 0; rc->arrived = 0; rc->off = 0;
  if (!have3 || !rc->cost_valid[slot3]) return;
  rc->cost_valid[slot3] = 0;
  const uint32_t c = rc->cost[slot3];
  unsigned long long r = ((unsigned long long)bits3 << 8) / (c ? c : 1u);
  if (r > (1u << 20)) r = 1u << 20;
  if (r < 1) r = 1;
  rc->ratio_q8 = rc->ratio_valid ? (uint32_t)((3ull * rc->ratio_q8 + r + 2) >> 2) : (uint32_t)r;
  rc->ratio_valid = 1;
}

// one workgroup per CTU of the group just reconstructed (f.row0, f.nrows); the last one to arrive decides for the next group
// (CTU rows [row0 + nrows, r2)), or, after the last group (r2 == row0 + nrows), files the picture's cost under `slot`
__global__ __launch_bounds__(256) void k_rc_band(EncFrame f, RcState *rc, long long T, int rows_total, int r2, int slot)
{
  __shared__ uint32_t part[4];
  __shared__ int last_s, off_s;
  const int tid = threadIdx.x, wc = f.cw >> 6, ctu = (int)blockIdx.x + f.row0 * wc, cx = ctu % wc, cy = ctu / wc;
  uint32_t c = 0;
  auto price = [&](uint32_t pair) {
    const int a = (int)(int16_t)(pair & 0xffffu), b = (int)pair >> 16;
    if (a) c += 3u + 2u * (uint32_t)(31 - __builtin_clz((unsigned)iabs(a)));
    if (b) c += 3u + 2u * (uint32_t)(31 - __builtin_clz((unsigned)iabs(b)));
  };
  {
    // (the level planes are only written where a block has levels: what lies under a block without coded levels is an older picture's)
    const int16_t *p = f.coef[0] + (size_t)(cy * 64) * f.cw + cx * 64;
    for (int i = tid; i < 64 * 8; i += 256) {                     // luma: 64 rows x 8 pieces of 8 levels
      const int y = i >> 3, xp = i & 7;
      if (!(f.cu_cbf[b8idx(f, cx * 64 + xp * 8, cy * 64 + y)] & 1)) continue;
      const uint4 v = *(const uint4 *)(p + (size_t)y * f.cw + xp * 8);
      price(v.x); price(v.y); price(v.z); price(v.w);
    }
    const int cw2 = f.cw >> 1;
    for (int i = tid; i < 2 * 32 * 4; i += 256) {                 // two chroma planes: 32 rows x 4 pieces
      const int pl = i >> 7, r = i & 127, y = r >> 2, xp = r & 3;
      if (!((f.cu_cbf[b8idx(f, cx * 64 + xp * 16, cy * 64 + y * 2)] >> (1 + pl)) & 1)) continue;
      const uint4 v = *(const uint4 *)(f.coef[1 + pl] + (size_t)(cy * 32 + y) * cw2 + cx * 32 + xp * 8);
      price(v.x); price(v.y); price(v.z); price(v.w);
    }
  }
  c = wave_sum_u32(c);
  if ((tid & 63) == 0) part[tid >> 6] = c;
  __syncthreads();
  if (tid == 0) {
    atomicAdd(&rc->cost_sofar, part[0] + part[1] + part[2] + part[3]);
    __threadfence();
    last_s = atomicAdd(&rc->arrived, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last_s) return;
  const int r1 = f.row0 + f.nrows;
  if (tid == 0) {
    const uint32_t total = atomicAdd(&rc->cost_sofar, 0u);
    rc->arrived = 0;
    int off = rc->off;
    if (r2 > r1) {
      if (rc->ratio_valid) {
        const unsigned long long est = ((unsigned long long)total * rc->ratio_q8) >> 8, tgt = ((unsigned long long)T * (unsigned long long)r1) / (unsigned long long)rows_total;
        if (est * 8 > tgt * 9) off++; else if (est * 8 < tgt * 7) off--;
        off = clip3(-3, 3, off);
      }
      rc->off = off;
    } else { rc->cost[slot] = total; rc->cost_valid[slot] = 1; }
    off_s = off;
  }
  __syncthreads();
  if (r2 > r1 && off_s) {
    int8_t *qt = const_cast<int8_t *>(f.ctu_qt);
    for (int i = r1 * wc + tid; i < r2 * wc; i += 256) qt[i] = (int8_t)clip3(0, 51, (int)qt[i] + off_s);
  }
}

void launch_rc_begin(RcState *rc, uint32_t bits3, int slot3, int have3, hipStream_t st) { hipLaunchKernelGGL(k_rc_begin, dim3(1), dim3(1), 0, st, rc, bits3, slot3, have3); }
void launch_rc_band(const EncFrame &f, RcState *rc, long long T, int rows_total, int r2, int slot, hipStream_t st)
{
  hipLaunchKernelGGL(k_rc_band, dim3((f.cw / 64) * f.nrows), dim3(256), 0, st, f, rc, T, rows_total, r2, slot);
}

}  // namespace kvzx
