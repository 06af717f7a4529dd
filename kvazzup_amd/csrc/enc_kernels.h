// kvazzup_amd/csrc/enc_kernels.h -- launch wrappers of the encoder kernels (enc_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "hevc_core.h"
namespace kvzx {
void launch_pad_input(const uint8_t *in, int w, int h, uint8_t *dst, int cw, int ch, hipStream_t st);
void launch_me(const EncFrame &f, hipStream_t st);
void launch_inter_recon(const EncFrame &f, hipStream_t st);
void launch_inter_signal(const EncFrame &f, hipStream_t st);
void launch_intra_analyse(const EncFrame &f, hipStream_t st);
void launch_intra_recon(const EncFrame &f, hipStream_t st);
void launch_deblock(const EncFrame &f, hipStream_t st);
void launch_entropy(const EncFrame &f, hipStream_t st);
}  // namespace kvzx
