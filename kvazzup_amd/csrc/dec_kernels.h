// kvazzup_amd/csrc/dec_kernels.h -- launch wrappers of the decoder kernels (dec_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "dec_frame.h"
namespace kvzx {
void launch_dec_inter(const DecFrame &f, hipStream_t st);     // prediction + residual of every inter block
void launch_dec_intra_resid(const DecFrame &f, hipStream_t st);   // residuals of the intra transform blocks up to 16x16, all at once (they do not depend on prediction)
void launch_dec_intra(const DecFrame &f, hipStream_t st);     // intra blocks, wavefront over CTUs (f.progress must be zero)
void launch_dec_deblock(const DecFrame &f, hipStream_t st);   // in place on f.rec
void launch_dec_sao(const DecFrame &f, hipStream_t st);       // f.rec -> f.out with the parameters in f.sao
// ---- the same kernels over up to KVZ_DEC_BATCH_MAX pictures of different decoders in one launch (batch.h).  h[i]: host copy of frame i (geometry for the
// grid), d[i]: its copy in device memory; frames a kernel has nothing to do for are passed as nullptr in d[] and get no workgroups.
void launch_dec_inter_n(const DecFrame *const *h, const DecFrame *const *d, int n, hipStream_t st);
void launch_dec_intra_n(const DecFrame *const *h, const DecFrame *const *d, int n, hipStream_t st);     // residuals + chain (two launches)
void launch_dec_deblock_n(const DecFrame *const *h, const DecFrame *const *d, int n, hipStream_t st);
void launch_dec_sao_n(const DecFrame *const *h, const DecFrame *const *d, int n, hipStream_t st);
}  // namespace kvzx
