// kvazzup_amd/csrc/enc_kernels.h -- launch wrappers of the encoder kernels (enc_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "hevc_core.h"
namespace kvzx {
void launch_pad_input(const uint8_t *in, int w, int h, uint8_t *dy, uint8_t *du, uint8_t *dv, int cw, int ch, hipStream_t st);   // packed I420 -> padded planes
void launch_me(const EncFrame &f, hipStream_t st);
void launch_subpel(const EncFrame &f, hipStream_t st);     // k_subpel: fractional-sample refinement of the searched vectors (f.subme > 0)
void launch_inter_recon(const EncFrame &f, hipStream_t st);
void launch_inter_signal(const EncFrame &f, hipStream_t st);
void launch_intra_analyse(const EncFrame &f, hipStream_t st);
void launch_intra_recon(const EncFrame &f, hipStream_t st);
void launch_deblock(const EncFrame &f, hipStream_t st);
void launch_vaq(const EncFrame &f, int vaq, int *act, int *sum, hipStream_t st);   // VAQ: f.ctu_qt holds ROI deltas on entry, target QPs on exit
void launch_sao(const EncFrame &f, hipStream_t st);        // SAO decision + filter: f.rec (deblocked) -> f.sao_out, parameters -> f.sao
void launch_qp_resolve(const EncFrame &f, hipStream_t st);  // per-CTU QP: first coded CU, QpY, delta (no-op without a QP map)
void launch_deblock_v(const EncFrame &f, hipStream_t st);   // vertical edges of the band
void launch_deblock_h(const EncFrame &f, hipStream_t st, int part = 0);   // part: 0 all horizontal edges of the band, 1 the inner ones, 2 its two boundary edges
void launch_tokenize(const EncFrame &f, hipStream_t st);    // k_tokenize: bins of every CTU into its slot, pieces in completion order
void launch_tok_compact(const EncFrame &f, hipStream_t st); // k_tok_compact: coding order restored, dense copy to host-mapped memory
// rate control v2 (rc_kernels.hip; the groups of CTU rows themselves: k_inter_recon's RC form, EncFrame::rc)
void launch_picture_begin(RcState *rc, uint32_t bits3, int slot3, int have3, int8_t *ctu_qt, const int8_t *roi, int nctu, int qp, int vaq, hipStream_t st,
                          void *zero_a = nullptr, size_t bytes_a = 0, void *zero_b = nullptr, size_t bytes_b = 0);      // head of a picture's chain: rate control state (rc != NULL) and the per-CTU target QPs (ctu_qt != NULL) in one launch
// k_cabac_rows (cabac_kernels.hip): the arithmetic coder proper on the GPU, one wave per substream
struct CabacRowsArgs {
  const uint16_t *tok; const int32_t *count; const uint32_t *off;   // dense tokens (device): CTU i has count[i] tokens at tok + off[i]
  uint8_t *stage; uint32_t stage_cap;     // device: every substream codes into a range reserved for its worst case
  uint8_t *out; uint32_t out_cap;         // host-mapped: the finished substreams, dense, in completion order
  uint32_t *cursors;                      // device {staging bytes reserved, output bytes reserved}: zero before the launch (k_tok_compact)
  uint32_t *sub_off, *sub_len, *sub_bins; // host-mapped, per substream of the launch: byte range in `out`, bins coded; sub_len == ~0u: did not fit
  uint32_t *ctx_save, *ctx_ready;         // device [CTU rows][40] context words after the row's second CTU, [CTU rows] generation of that copy
  uint32_t gen;                           // generation of this launch (differs from the previous launch with the same arrays)
  uint32_t *err;                          // device error flags (64: a row waited for its upper neighbour in vain)
  int wc, hc, wpp, tile_rows, init_type, qp, first_sub;
};
void launch_cabac_rows(const CabacRowsArgs &a, int nsub, hipStream_t st);
void launch_cabac_decode_probe(const CabacRowsArgs &a, int nsub, uint32_t *mismatch, hipStream_t st);      // measurement aid (cabac_kernels.hip)
}  // namespace kvzx
