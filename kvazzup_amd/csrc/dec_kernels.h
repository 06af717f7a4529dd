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
}  // namespace kvzx
