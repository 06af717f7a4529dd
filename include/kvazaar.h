/*
 * include/kvazaar.h -- C ABI of the MI355X-native HEVC encoder, source-compatible with the part
 * of Kvazaar's public header that uvgComm compiles against.
 *
 * uvgComm reaches its encoder only through this interface
 * (/root/reference/src/media/processing/kvazaarfilter.cpp:8 `#include <kvazaar.h>`):
 *   kvz_api_get(8)                          kvazaarfilter.cpp:145
 *   config_alloc / config_init              kvazaarfilter.cpp:151,160
 *   config_parse(cfg, name, value) == 1 ok  kvazaarfilter.cpp:172-287,363-367
 *   fields written directly                 kvazaarfilter.cpp:223 (target_bitrate), :244 (lossless),
 *                                           :257-276 (mv_constraint), :278 (set_qp_in_cu), :289 (hash)
 *   fields read directly                    kvazaarfilter.cpp:207 (wpp), :299 (owf), :381-384 (width,
 *                                           height, framerate_num, framerate_denom)
 *   encoder_open / encoder_close            kvazaarfilter.cpp:291,317
 *   config_destroy                          kvazaarfilter.cpp:318
 *   picture_alloc / picture_free            kvazaarfilter.cpp:69,54,476
 *   kvz_picture.{y,u,v,pts,roi}             kvazaarfilter.cpp:410-430,53
 *   encoder_encode                          kvazaarfilter.cpp:435-438,445-448
 *   kvz_data_chunk.{data,len,next}          kvazaarfilter.cpp:469-474
 *   chunk_free                              kvazaarfilter.cpp:475
 *
 * The Kvazaar 2.3.1 header itself is not in /root/reference (FetchContent dependency,
 * dependencies/kvazaar.cmake:10-14); the declarations below keep its names and meanings so that
 * kvazaarfilter.cpp compiles unchanged against this file and links against libkvazzup_amd.so
 * (see INTEGRATION.md).  Binary layout compatibility with a Kvazaar-built libkvazaar is NOT
 * claimed: the application must be compiled against this header.
 */
#ifndef KVAZZUP_AMD_KVAZAAR_H_
#define KVAZZUP_AMD_KVAZAAR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define KVZ_PUBLIC __declspec(dllexport)
#else
#define KVZ_PUBLIC __attribute__((visibility("default")))
#endif

#define KVZ_BIT_DEPTH 8
#define KVZ_DATA_CHUNK_SIZE 4096
#define KVZ_MAX_GOP_LENGTH 32

typedef uint8_t kvz_pixel;
typedef struct kvz_encoder kvz_encoder;

enum kvz_chroma_format { KVZ_CSP_400 = 0, KVZ_CSP_420 = 1, KVZ_CSP_422 = 2, KVZ_CSP_444 = 3 };
enum kvz_interlacing { KVZ_INTERLACING_NONE = 0, KVZ_INTERLACING_TFF = 1, KVZ_INTERLACING_BFF = 2 };
enum kvz_mv_constraint {
  KVZ_MV_CONSTRAIN_NONE = 0,
  KVZ_MV_CONSTRAIN_FRAME = 1,
  KVZ_MV_CONSTRAIN_TILE = 2,
  KVZ_MV_CONSTRAIN_FRAME_AND_TILE = 3,
  KVZ_MV_CONSTRAIN_FRAME_AND_TILE_MARGIN = 4
};
enum kvz_hash { KVZ_HASH_NONE = 0, KVZ_HASH_CHECKSUM = 1, KVZ_HASH_MD5 = 2 };
enum kvz_slices { KVZ_SLICES_NONE = 0, KVZ_SLICES_TILES = 1, KVZ_SLICES_WPP = 2 };
enum kvz_sao { KVZ_SAO_OFF = 0, KVZ_SAO_EDGE = 1, KVZ_SAO_BAND = 2, KVZ_SAO_FULL = 3 };
enum kvz_scalinglist { KVZ_SCALING_LIST_OFF = 0, KVZ_SCALING_LIST_CUSTOM = 1, KVZ_SCALING_LIST_DEFAULT = 2 };
enum kvz_rc_algorithm { KVZ_NO_RC = 0, KVZ_LAMBDA = 1, KVZ_OBA = 2 };
enum kvz_ime_algorithm { KVZ_IME_HEXBS = 0, KVZ_IME_TZ = 1, KVZ_IME_FULL = 2, KVZ_IME_FULL8 = 3,
                         KVZ_IME_FULL16 = 4, KVZ_IME_FULL32 = 5, KVZ_IME_FULL64 = 6, KVZ_IME_DIA = 7 };
enum kvz_nal_unit_type { KVZ_NAL_TRAIL_N = 0, KVZ_NAL_TRAIL_R = 1, KVZ_NAL_IDR_W_RADL = 19, KVZ_NAL_IDR_N_LP = 20,
                         KVZ_NAL_CRA_NUT = 21, KVZ_NAL_VPS_NUT = 32, KVZ_NAL_SPS_NUT = 33, KVZ_NAL_PPS_NUT = 34 };
enum kvz_slice_type { KVZ_SLICE_B = 0, KVZ_SLICE_P = 1, KVZ_SLICE_I = 2 };

/* Encoder configuration.  Fields uvgComm touches keep Kvazaar's names; the rest record what
 * config_parse() understood.  Initialise with kvz_api.config_init(). */
typedef struct kvz_config {
  int32_t qp;                 /* "qp" */
  int32_t intra_period;       /* "period": 0 = only first picture, 1 = all intra, n = every n-th */
  int32_t vps_period;         /* "vps-period": parameter sets with every n-th intra picture (0 = first only) */
  int32_t width;              /* "input-res" */
  int32_t height;
  double framerate;           /* deprecated in Kvazaar; kept for source compatibility */
  int32_t framerate_num;      /* "input-fps" */
  int32_t framerate_denom;
  int32_t deblock_enable;     /* "deblock" */
  enum kvz_sao sao_type;      /* "sao": off, or full (edge and band offsets) */
  int32_t rdoq_enable, signhide_enable;   /* "rdoq", "signhide": rate-distortion optimised quantisation ("uvgx RDOQ v1") and sign data hiding in the quantiser (oracle/hevc_enc.c) */
  int32_t smp_enable, amp_enable;         /* recorded; config_parse rejects the value 1 (2NxN / Nx2N / AMP inter partitions are not searched) */
  int32_t rdo;                /* "rd" */
  int32_t full_intra_search, trskip_enable, tr_depth_intra;
  enum kvz_ime_algorithm ime_algorithm;   /* "me": recorded; the GPU search is always exhaustive over +-me_range (a superset of what any of Kvazaar's patterns visits) */
  int32_t fme_level;          /* "subme" (0: integer samples only) */
  int32_t bipred;
  int32_t deblock_beta, deblock_tc;
  int32_t ref_frames;         /* "ref" */
  int32_t tiles_width_count, tiles_height_count;   /* "tiles" CxR: C tile columns x R tile rows, uniform spacing, at most 20 x 22 (a grid finer than the CTU grid is cut down to it) */
  int32_t wpp;                /* "wpp" */
  int32_t owf;                /* "owf": pictures in flight; output is delayed by this many calls */
  int32_t slices;             /* "slices": enum kvz_slices: wpp = a dependent slice segment per CTU row (needs wpp), tiles = an independent slice per tile; one NAL unit each */
  int32_t threads;            /* "threads": host threads of the arithmetic-coding stage (-1 / "auto": 16; 0: the calling thread only) */
  int32_t cpuid;
  int32_t lossless;           /* must be 0 */
  int32_t tmvp_enable;
  int32_t rdoq_skip, implicit_rdpcm;
  int32_t mv_rdo;
  int32_t calc_psnr;
  enum kvz_mv_constraint mv_constraint;
  enum kvz_hash hash;         /* "hash": decoded picture hash SEI after every picture: none, checksum or md5 (D.3.19), computed from the reconstruction on the device */
  int32_t cu_split_termination, me_early_termination, intra_rdo_et, early_skip;
  int32_t target_bitrate;     /* "bitrate" in bits per second: 0 = constant QP; > 0 = this library's rate control (picture level; with rc_algorithm lambda / oba also between groups of CTU rows, decided on the device) */
  enum kvz_rc_algorithm rc_algorithm;
  int32_t max_merge;
  int32_t gop_len, gop_lowdelay;      /* "gop": lp-g<len>d<depth>t<layers> accepted; one reference is used */
  int32_t gop_lp_ref_depth, gop_lp_temporal_layers;
  int32_t set_qp_in_cu;
  int32_t vaq;
  enum kvz_scalinglist scaling_list;
  int32_t intra_bits;
  int32_t me_max_steps;
  int32_t fast_residual_cost_limit;
  int32_t pu_depth_inter_min, pu_depth_inter_max, pu_depth_intra_min, pu_depth_intra_max;
  /* kvazzup_amd extensions (not in Kvazaar) */
  int32_t me_range;           /* "me-range": exhaustive search radius in integer samples, 1..32 */
  int32_t gpu_device;         /* "gpu": HIP device ordinal */
  int32_t recon_output;       /* "recon-output": 0 = encoder_encode leaves *pic_out NULL (no download) */
  int32_t intra_satd;         /* "intra-satd": 1 (default) = the intra mode search compares 8x8 Hadamard sums (SATD) like Kvazaar's rough search, 0 = SAD */
  int32_t band_row0, band_rows; /* "band-row0", "band-rows": tile-row split over several encoders (kvazzup_amd.h, kvzx_encoder_band_*); 0 rows = whole picture */
  int32_t input_hold;         /* "input-hold": 1 = the caller leaves a DEVICE input picture (kvzx_encoder_encode_device) unchanged until that picture's access unit has come back, as the kvz_picture contract demands of host pictures; the call then returns without waiting for the input stage (0, default: the buffer may be reused when the call returns) */
  int32_t null_input_poll;    /* "null-input": "drain" (default, Kvazaar's meaning: encoder_encode with pic_in == NULL waits for the oldest picture in flight) or "poll" (it returns a picture only if one has already been finished and never waits: what the loop at kvazaarfilter.cpp:440-448 needs to keep video/OWF pictures in flight instead of emptying the pipeline after every picture) */
  int32_t intra_in_p;         /* "intra-in-p" 0 / 1 / 2: intra coding units in P pictures, 1 = 16x16 units only (presets superfast .. fast), 2 = 16x16 and 8x8 units (medium and slower) ("uvgx intra-in-P v1": a 16x16 quarter whose motion-search cost is high is priced as an intra block from the source picture and coded intra when that is cheaper -- scene cuts, uncovered background); under rate control v2 the row groups are priced without the intra units' levels; ignored in band mode */
  int32_t gpu_entropy;        /* "gpu-entropy": 1 = the arithmetic coder runs on the GPU too (k_cabac_rows: no host coder threads, 2-4 ms more latency per picture), 0 (default) = host thread pool sized by "threads" */
  int32_t intra_chain;        /* "intra-chain" (default 1): the blocks of a CTU whose below-left / above-right reference samples lie in ANOTHER CTU (its left-edge blocks, its above-right corner block) choose among the intra modes that do not read those samples: the CTU wavefront of the reconstruction chain (k_intra_recon, and k_dec_intra on the receiving side) advances in shorter lags; 0 = all 35 modes everywhere */
  int32_t me_source;          /* "me-source" 0 / 1 ("uvgx search pipelining v1"): the integer motion search of a P picture looks at the previous INPUT picture instead of the reference picture's reconstruction, so it depends on nothing the previous picture's reconstruction loop produces and runs beside it on the GPU (fractional refinement, motion compensation and everything behind them use the reconstruction as ever).  On at the presets superfast .. fast, whose subme >= 2 refinement against the reconstruction makes up for it (oracle, 640x384 / 720p: -0.6 .. +0.2 % bits, -0.01 .. -0.02 dB); off at ultrafast (no refinement there: +1.1 .. 1.4 % bits, -0.17 dB) and from medium on; ignored in band mode */
} kvz_config;

/* Picture.  y/u/v are planar 8-bit with stride == width (chroma width/2), as uvgComm assumes
 * (kvazaarfilter.cpp:410-418). */
typedef struct kvz_picture {
  kvz_pixel *fulldata_buf;    /* allocation holding all planes */
  kvz_pixel *fulldata;
  kvz_pixel *y, *u, *v;
  kvz_pixel *data[3];
  int32_t width, height, stride;
  struct kvz_picture *base_image;
  int32_t refcount;
  int64_t pts, dts;
  enum kvz_interlacing interlacing;
  enum kvz_chroma_format chroma_format;
  int32_t ref_pocs[16];
  struct {
    int width, height;        /* in 64x64 CTUs */
    int8_t *roi_array;        /* delta QP per CTU, raster; owned by the caller (kvazaarfilter.cpp:426-430,459-463) */
  } roi;
} kvz_picture;

typedef struct kvz_frame_info {
  int32_t poc;
  int8_t qp;
  enum kvz_nal_unit_type nal_unit_type;
  enum kvz_slice_type slice_type;
  int ref_list[2][16];
  int ref_list_len[2];
} kvz_frame_info;

typedef struct kvz_data_chunk {
  uint8_t data[KVZ_DATA_CHUNK_SIZE];
  uint32_t len;
  struct kvz_data_chunk *next;
} kvz_data_chunk;

typedef struct kvz_api {
  kvz_config *(*config_alloc)(void);
  int (*config_destroy)(kvz_config *cfg);
  int (*config_init)(kvz_config *cfg);
  int (*config_parse)(kvz_config *cfg, const char *name, const char *value);   /* 1 = accepted, 0 = rejected */

  kvz_picture *(*picture_alloc)(int32_t width, int32_t height);
  void (*picture_free)(kvz_picture *pic);                                      /* NULL is allowed */

  void (*chunk_free)(kvz_data_chunk *chunk);                                   /* frees the whole list */

  kvz_encoder *(*encoder_open)(const kvz_config *cfg);                         /* NULL on failure */
  void (*encoder_close)(kvz_encoder *encoder);
  int (*encoder_headers)(kvz_encoder *encoder, kvz_data_chunk **data_out, uint32_t *len_out);
  /* Feed one picture (or NULL to flush).  On return *data_out is the access unit of the oldest
   * finished picture (NULL if none), *len_out its size, *pic_out its reconstruction (refcounted,
   * release with picture_free), *src_out the matching source, *info_out its description.
   * Returns 1 on success, 0 on failure. */
  int (*encoder_encode)(kvz_encoder *encoder, kvz_picture *pic_in, kvz_data_chunk **data_out, uint32_t *len_out,
                        kvz_picture **pic_out, kvz_picture **src_out, kvz_frame_info *info_out);
  kvz_picture *(*picture_alloc_csp)(enum kvz_chroma_format chroma_format, int32_t width, int32_t height);
} kvz_api;

/* bit_depth must be 8 (NULL otherwise).  Getting the table does not touch the GPU; encoder_open()
 * returns NULL when no HIP device is usable -- there is no CPU fallback. */
KVZ_PUBLIC const kvz_api *kvz_api_get(int bit_depth);

#ifdef __cplusplus
}
#endif
#endif
