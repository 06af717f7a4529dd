"""ctypes view of libkvazzup_amd.so: the kvz_api table (include/kvazaar.h), the libOpenHevc*
functions (include/openHevcWrapper.h) and the extensions (include/kvazzup_amd.h)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def library_path():
    # (KVAZZUP_AMD_LIBRARY: an instrumented build of the same sources, tools/intra_prof.py)
    return os.environ.get("KVAZZUP_AMD_LIBRARY") or os.path.join(_HERE, "libkvazzup_amd.so")


def build_library(force=False):
    """Compile every HIP source for gfx950 into kvazzup_amd/libkvazzup_amd.so (in tree)."""
    args = ["make", "-s", "-j8", "-C", os.path.join(_HERE, "csrc")]
    if force:
        subprocess.run(args + ["clean"], check=True)
    subprocess.run(args, check=True)
    return library_path()


class KvzConfig(C.Structure):
    _fields_ = [("qp", C.c_int32), ("intra_period", C.c_int32), ("vps_period", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("framerate", C.c_double), ("framerate_num", C.c_int32), ("framerate_denom", C.c_int32), ("deblock_enable", C.c_int32),
                ("sao_type", C.c_int), ("rdoq_enable", C.c_int32), ("signhide_enable", C.c_int32), ("smp_enable", C.c_int32), ("amp_enable", C.c_int32),
                ("rdo", C.c_int32), ("full_intra_search", C.c_int32), ("trskip_enable", C.c_int32), ("tr_depth_intra", C.c_int32),
                ("ime_algorithm", C.c_int), ("fme_level", C.c_int32), ("bipred", C.c_int32), ("deblock_beta", C.c_int32), ("deblock_tc", C.c_int32),
                ("ref_frames", C.c_int32), ("tiles_width_count", C.c_int32), ("tiles_height_count", C.c_int32), ("wpp", C.c_int32), ("owf", C.c_int32),
                ("slices", C.c_int32), ("threads", C.c_int32), ("cpuid", C.c_int32), ("lossless", C.c_int32), ("tmvp_enable", C.c_int32),
                ("rdoq_skip", C.c_int32), ("implicit_rdpcm", C.c_int32), ("mv_rdo", C.c_int32), ("calc_psnr", C.c_int32),
                ("mv_constraint", C.c_int), ("hash", C.c_int), ("cu_split_termination", C.c_int32), ("me_early_termination", C.c_int32),
                ("intra_rdo_et", C.c_int32), ("early_skip", C.c_int32), ("target_bitrate", C.c_int32), ("rc_algorithm", C.c_int),
                ("max_merge", C.c_int32), ("gop_len", C.c_int32), ("gop_lowdelay", C.c_int32), ("gop_lp_ref_depth", C.c_int32),
                ("gop_lp_temporal_layers", C.c_int32), ("set_qp_in_cu", C.c_int32), ("vaq", C.c_int32), ("scaling_list", C.c_int),
                ("intra_bits", C.c_int32), ("me_max_steps", C.c_int32), ("fast_residual_cost_limit", C.c_int32),
                ("pu_depth_inter_min", C.c_int32), ("pu_depth_inter_max", C.c_int32), ("pu_depth_intra_min", C.c_int32), ("pu_depth_intra_max", C.c_int32),
                ("me_range", C.c_int32), ("gpu_device", C.c_int32), ("recon_output", C.c_int32), ("intra_satd", C.c_int32),
                ("band_row0", C.c_int32), ("band_rows", C.c_int32), ("input_hold", C.c_int32), ("null_input_poll", C.c_int32), ("intra_in_p", C.c_int32), ("gpu_entropy", C.c_int32), ("intra_chain", C.c_int32), ("me_source", C.c_int32)]


class KvzRoi(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("roi_array", C.POINTER(C.c_int8))]


class KvzPicture(C.Structure):
    pass


KvzPicture._fields_ = [("fulldata_buf", C.c_void_p), ("fulldata", C.c_void_p), ("y", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p),
                       ("data", C.c_void_p * 3), ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32),
                       ("base_image", C.POINTER(KvzPicture)), ("refcount", C.c_int32), ("pts", C.c_int64), ("dts", C.c_int64),
                       ("interlacing", C.c_int), ("chroma_format", C.c_int), ("ref_pocs", C.c_int32 * 16), ("roi", KvzRoi)]


class KvzFrameInfo(C.Structure):
    _fields_ = [("poc", C.c_int32), ("qp", C.c_int8), ("nal_unit_type", C.c_int), ("slice_type", C.c_int),
                ("ref_list", (C.c_int * 16) * 2), ("ref_list_len", C.c_int * 2)]


class KvzDataChunk(C.Structure):
    pass


KvzDataChunk._fields_ = [("data", C.c_uint8 * 4096), ("len", C.c_uint32), ("next", C.POINTER(KvzDataChunk))]

_P = C.POINTER


class KvzApi(C.Structure):
    _fields_ = [("config_alloc", C.CFUNCTYPE(_P(KvzConfig))),
                ("config_destroy", C.CFUNCTYPE(C.c_int, _P(KvzConfig))),
                ("config_init", C.CFUNCTYPE(C.c_int, _P(KvzConfig))),
                ("config_parse", C.CFUNCTYPE(C.c_int, _P(KvzConfig), C.c_char_p, C.c_char_p)),
                ("picture_alloc", C.CFUNCTYPE(_P(KvzPicture), C.c_int32, C.c_int32)),
                ("picture_free", C.CFUNCTYPE(None, _P(KvzPicture))),
                ("chunk_free", C.CFUNCTYPE(None, _P(KvzDataChunk))),
                ("encoder_open", C.CFUNCTYPE(C.c_void_p, _P(KvzConfig))),
                ("encoder_close", C.CFUNCTYPE(None, C.c_void_p)),
                ("encoder_headers", C.CFUNCTYPE(C.c_int, C.c_void_p, _P(_P(KvzDataChunk)), _P(C.c_uint32))),
                ("encoder_encode", C.CFUNCTYPE(C.c_int, C.c_void_p, _P(KvzPicture), _P(_P(KvzDataChunk)), _P(C.c_uint32),
                                               _P(_P(KvzPicture)), _P(_P(KvzPicture)), _P(KvzFrameInfo))),
                ("picture_alloc_csp", C.CFUNCTYPE(_P(KvzPicture), C.c_int, C.c_int32, C.c_int32))]


class OpenHevcRational(C.Structure):
    _fields_ = [("num", C.c_int), ("den", C.c_int)]


class OpenHevcFrameInfo(C.Structure):
    _fields_ = [("nYPitch", C.c_int), ("nUPitch", C.c_int), ("nVPitch", C.c_int), ("nBitDepth", C.c_int), ("nWidth", C.c_int),
                ("nHeight", C.c_int), ("chromat_format", C.c_int), ("sample_aspect_ratio", OpenHevcRational),
                ("frameRate", OpenHevcRational), ("display_picture_number", C.c_int), ("flag", C.c_int), ("nTimeStamp", C.c_int64)]


class OpenHevcFrame(C.Structure):
    _fields_ = [("pvY", C.c_void_p), ("pvU", C.c_void_p), ("pvV", C.c_void_p), ("frameInfo", OpenHevcFrameInfo)]


ENCODER_EXPORTS = ["kvz_api_get", "kvzx_version", "kvzx_rgb32_to_yuv420", "kvzx_rgb32_to_yuv420_device", "kvzx_yuv420_to_rgb32", "kvzx_yuv420_to_rgb32_device", "uvgx_pipeline_flush", "kvzx_device_count", "kvzx_encoder_encode_device", "kvzx_encoder_encode_host",
                   "kvzx_encoder_coded_size", "kvzx_encoder_download_recon", "kvzx_encoder_recon_device", "kvzx_encoder_debug_copy",
                   "kvzx_encoder_set_profiling", "kvzx_encoder_kernel_times", "kvzx_encoder_kernel_name", "kvzx_encoder_last_bins", "kvzx_encoder_pending"]
DECODER_EXPORTS = ["libOpenHevcInit", "libOpenHevcStartDecoder", "libOpenHevcDecode", "libOpenHevcGetPictureInfo",
                   "libOpenHevcGetPictureSize2", "libOpenHevcGetOutput", "libOpenHevcGetOutputCpy", "libOpenHevcSetCheckMD5",
                   "libOpenHevcSetDebugMode", "libOpenHevcSetTemporalLayer_id", "libOpenHevcSetNoCropping", "libOpenHevcSetActiveDecoders",
                   "libOpenHevcSetViewLayers", "libOpenHevcClose", "libOpenHevcFlush", "libOpenHevcVersion",
                   "kvzx_decoder_set_device", "kvzx_decoder_last_error", "kvzx_decoder_hash_stats", "kvzx_decoder_output_device", "kvzx_decoder_set_download", "kvzx_decoder_set_output_hold", "kvzx_decoder_set_profiling",
                   "kvzx_decoder_kernel_times", "kvzx_decoder_kernel_name", "kvzx_decoder_debug_copy",
                   "kvzx_decoder_set_band", "kvzx_decoder_band_halo_bytes", "kvzx_decoder_band_export", "kvzx_decoder_band_import", "kvzx_decoder_band_deblock", "kvzx_decoder_band_finish", "kvzx_decoder_band_ready"]


def load_library():
    """Load libkvazzup_amd.so.  Raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError("kvazzup_amd: %s is missing -- run __graft_entry__.build() (hipcc, gfx950); there is no CPU fallback" % path)
    L = C.CDLL(path)
    L.kvz_api_get.restype = _P(KvzApi)
    L.kvz_api_get.argtypes = [C.c_int]
    L.kvzx_version.restype = C.c_char_p
    L.kvzx_encoder_encode_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, _P(C.c_uint32), _P(KvzFrameInfo)]
    L.kvzx_encoder_encode_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, _P(C.c_uint32), _P(KvzFrameInfo)]
    L.kvzx_encoder_coded_size.argtypes = [C.c_void_p, _P(C.c_int), _P(C.c_int)]
    L.kvzx_encoder_download_recon.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.kvzx_encoder_recon_device.argtypes = [C.c_void_p, _P(C.c_void_p)]
    L.kvzx_encoder_debug_copy.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    L.kvzx_encoder_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.kvzx_encoder_kernel_times.argtypes = [C.c_void_p, _P(C.c_double), _P(C.c_uint64), C.c_int]
    L.kvzx_encoder_kernel_name.restype = C.c_char_p
    L.kvzx_encoder_kernel_name.argtypes = [C.c_int]
    L.kvzx_encoder_last_bins.restype = C.c_uint64
    L.kvzx_encoder_last_bins.argtypes = [C.c_void_p]
    if hasattr(L, "libOpenHevcInit"):
        L.libOpenHevcInit.restype = C.c_void_p
        L.libOpenHevcInit.argtypes = [C.c_int, C.c_int]
        L.libOpenHevcStartDecoder.argtypes = [C.c_void_p]
        L.libOpenHevcDecode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64]
        L.libOpenHevcGetPictureInfo.argtypes = [C.c_void_p, _P(OpenHevcFrameInfo)]
        L.libOpenHevcGetPictureSize2.argtypes = [C.c_void_p, _P(OpenHevcFrameInfo)]
        L.libOpenHevcGetOutput.argtypes = [C.c_void_p, C.c_int, _P(OpenHevcFrame)]
        L.libOpenHevcGetOutputCpy.argtypes = [C.c_void_p, C.c_int, _P(OpenHevcFrame)]
        for n in ("libOpenHevcSetCheckMD5", "libOpenHevcSetDebugMode", "libOpenHevcSetTemporalLayer_id", "libOpenHevcSetNoCropping",
                  "libOpenHevcSetActiveDecoders", "libOpenHevcSetViewLayers"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_int]
        L.libOpenHevcClose.argtypes = [C.c_void_p]
        L.libOpenHevcFlush.argtypes = [C.c_void_p]
        L.libOpenHevcVersion.restype = C.c_char_p
        L.libOpenHevcVersion.argtypes = [C.c_void_p]
        L.kvzx_decoder_set_device.argtypes = [C.c_void_p, C.c_int]
        L.kvzx_decoder_last_error.argtypes = [C.c_void_p]
        L.kvzx_decoder_hash_stats.argtypes = [C.c_void_p, _P(C.c_int), _P(C.c_int)]
        L.kvzx_decoder_hash_stats.restype = None
        L.kvzx_decoder_output_device.argtypes = [C.c_void_p, _P(C.c_void_p), _P(C.c_int)]
        L.kvzx_decoder_set_download.argtypes = [C.c_void_p, C.c_int]
        L.kvzx_decoder_set_band.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.kvzx_decoder_band_halo_bytes.restype = C.c_size_t
        L.kvzx_decoder_band_halo_bytes.argtypes = [C.c_void_p]
        L.kvzx_decoder_band_export.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.kvzx_decoder_band_import.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.kvzx_decoder_band_deblock.argtypes = [C.c_void_p]
        L.kvzx_decoder_band_finish.argtypes = [C.c_void_p]
        L.kvzx_decoder_band_ready.argtypes = [C.c_void_p]
        L.kvzx_decoder_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.kvzx_decoder_kernel_times.argtypes = [C.c_void_p, _P(C.c_double), _P(C.c_uint64), C.c_int]
        L.kvzx_decoder_kernel_name.restype = C.c_char_p
        L.kvzx_decoder_kernel_name.argtypes = [C.c_int]
        L.kvzx_decoder_debug_copy.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    _LIB = L
    return L
