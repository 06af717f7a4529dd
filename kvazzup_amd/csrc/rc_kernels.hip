// kvazzup_amd/csrc/rc_kernels.hip -- "uvgx rate control v2": feedback inside the picture, decided on the device.
//
// The picture-level controller (Encoder::rate_control, statement rate_control() in oracle/hevc_enc.c) sets the picture's QP from
// access-unit sizes a few pictures old.  On top of it a P picture's CTU rows are reconstructed in a few groups: k_inter_recon
// (enc_kernels.hip, the RC form) prices the levels of a group as it quantises them -- level cost of a block = sum over its non-zero
// levels of 3 + 2 floor(log2 |level|), one unit worth ratio / 256 bits as measured on earlier P pictures -- against the share of the
// picture's target the rows done are entitled to, and the group's last workgroup moves the QP of the next group's CTUs one step
// (within +-3 of the picture's QP; rc_group_done() there).  All groups are ONE launch: the next group's workgroups are dispatched
// behind this one's, predict and transform, and wait for the decision in front of their quantiser (rounds 2 and early 3: one launch
// per group with a pricing kernel in between, 4 x (19 + 4) us per 1080p picture).  The host never sees these QPs: they go into the
// per-CTU QP array the quantiser, the cu_qp_delta chain (k_qp_first / k_qp_chain) and deblocking read anyway.  Statement of
// record: rc_ctu_cost(), rc_band_decide(), rc_picture_start() in oracle/hevc_enc.c.
#include <hip/hip_runtime.h>
#include "hevc_core.h"
#include "enc_kernels.h"
#include "kernel_common.h"

namespace kvzx {

// The head of a picture's chain on the main stream, ONE launch (round 3: a one-thread kernel for the rate control state plus a host-to-device copy of the
// per-CTU target QPs queued on the main stream -- 5 us + a ~28 us bubble in front of every picture of uvgComm's default mode):
//   * rate control v2 (rc != NULL): the groups' counters back to zero; the access unit of a few pictures ago has been sized (bits3), and if that picture was
//     coded in groups its bits per unit of level cost update the ratio;
//   * cu_qp_delta (ctu_qt != NULL): every CTU's target QP = clip(picture QP + ROI delta) -- with VAQ the ROI delta alone, k_vaq_apply adds the rest.  The ROI
//     deltas (per CTU, already spread over the CTU grid) were uploaded on the input stream when the picture brought a map; NULL = no map.
//   * an intra picture: the two arrays its chain starts from (the ticket counter's array, the CU cbf bits the three plane waves OR into) go back to zero
//     here -- two memsets of their own were two more launches in front of every intra picture's chain
__global__ __launch_bounds__(256) void k_picture_begin(RcState *rc, uint32_t bits3, int slot3, int have3, int8_t *ctu_qt, const int8_t *roi, int nctu, int qp, int vaq,
                                                       uint4 *zero_a, int na16, uint4 *zero_b, int nb16)
{
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (int i = threadIdx.x; i < na16; i += 256) zero_a[i] = z;
  for (int i = threadIdx.x; i < nb16; i += 256) zero_b[i] = z;
  picture_begin_body(rc, bits3, slot3, have3, ctu_qt, roi, nctu, qp, vaq, (int)threadIdx.x, 256);
}
void launch_picture_begin(RcState *rc, uint32_t bits3, int slot3, int have3, int8_t *ctu_qt, const int8_t *roi, int nctu, int qp, int vaq, hipStream_t st,
                          void *zero_a, size_t bytes_a, void *zero_b, size_t bytes_b)
{
  // (bytes_a / bytes_b: multiples of 16 -- the caller rounds its arrays' sizes up when it allocates them)
  if (rc || ctu_qt || bytes_a || bytes_b)
    hipLaunchKernelGGL(k_picture_begin, dim3(1), dim3(256), 0, st, rc, bits3, slot3, have3, ctu_qt, roi, nctu, qp, vaq, (uint4 *)zero_a, (int)(bytes_a / 16), (uint4 *)zero_b, (int)(bytes_b / 16));
}
}  // namespace kvzx
