/*
 * include/openHevcWrapper.h -- C ABI of the MI355X-native HEVC decoder, source-compatible with
 * the OpenHEVC wrapper header uvgComm compiles against
 * (/root/reference/src/media/processing/openhevcfilter.h:4 `#include "openHevcWrapper.h"`).
 *
 * Call sites being served:
 *   libOpenHevcInit(threads, type)          openhevcfilter.cpp:36-47   (type: 1 frame, 2 slice, 3 both)
 *   libOpenHevcStartDecoder                 openhevcfilter.cpp:49      (-1 = failure)
 *   libOpenHevcSetTemporalLayer_id / SetActiveDecoders / SetViewLayers   openhevcfilter.cpp:54-56
 *       (SetTemporalLayer_id: the highest temporal sub-layer decoded, OpenHEVC's default 7 = all -- slice NAL units above it are dropped; uvgComm passes 0.
 *        SetActiveDecoders / SetViewLayers: layered extensions, no effect here -- NAL units of nuh_layer_id > 0 are dropped.  SetNoCropping: pictures at their
 *        coded size.  SetCheckMD5: decoded picture hash SEI messages are compared.  SetDebugMode: no effect.)
 *   libOpenHevcVersion                      openhevcfilter.cpp:64
 *   libOpenHevcDecode(h, nal, len, pts)     openhevcfilter.cpp:145-146 (<0 error, 0 none, >0 picture)
 *   libOpenHevcGetOutput(h, got, &frame)    openhevcfilter.cpp:195     (>0 = frame filled)
 *   libOpenHevcGetPictureInfo(h, &info)     openhevcfilter.cpp:199
 *   frame.pvY/pvU/pvV + frameInfo.nYPitch/nUPitch/nWidth/nHeight/frameRate   openhevcfilter.cpp:201-233
 *   libOpenHevcFlush / libOpenHevcClose     openhevcfilter.cpp:81-82
 *
 * The OpenHEVC header is not in /root/reference (fetched by dependencies/openhevc.cmake:10-25);
 * names and meanings below follow its public interface as uvgComm uses it.  One NAL unit per
 * libOpenHevcDecode call, Annex-B start code included (uvgrtpreceiver.cpp:86-112).  Frame
 * memory returned by libOpenHevcGetOutput belongs to the decoder and stays valid until the next
 * libOpenHevcDecode call.
 */
#ifndef KVAZZUP_AMD_OPENHEVCWRAPPER_H_
#define KVAZZUP_AMD_OPENHEVCWRAPPER_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define OHEVC_PUBLIC __declspec(dllexport)
#else
#define OHEVC_PUBLIC __attribute__((visibility("default")))
#endif

typedef void *OpenHevc_Handle;

typedef struct OpenHevc_Rational { int num; int den; } OpenHevc_Rational;

enum ChromaFormat { YUV420 = 0, YUV422, YUV444 };

typedef struct OpenHevc_FrameInfo {
  int nYPitch;
  int nUPitch;
  int nVPitch;
  int nBitDepth;
  int nWidth;
  int nHeight;
  int chromat_format;
  OpenHevc_Rational sample_aspect_ratio;
  OpenHevc_Rational frameRate;
  int display_picture_number;
  int flag;
  int64_t nTimeStamp;
} OpenHevc_FrameInfo;

typedef struct OpenHevc_Frame {
  void **pvY;
  void **pvU;
  void **pvV;
  OpenHevc_FrameInfo frameInfo;
} OpenHevc_Frame;

typedef struct OpenHevc_Frame_cpy {
  void *pvY;
  void *pvU;
  void *pvV;
  OpenHevc_FrameInfo frameInfo;
} OpenHevc_Frame_cpy;

OHEVC_PUBLIC OpenHevc_Handle libOpenHevcInit(int nb_pthreads, int thread_type);
OHEVC_PUBLIC int libOpenHevcStartDecoder(OpenHevc_Handle h);                     /* -1 when no HIP device is usable */
OHEVC_PUBLIC int libOpenHevcDecode(OpenHevc_Handle h, const unsigned char *buff, int nal_len, int64_t pts);
OHEVC_PUBLIC void libOpenHevcGetPictureInfo(OpenHevc_Handle h, OpenHevc_FrameInfo *info);
OHEVC_PUBLIC void libOpenHevcGetPictureSize2(OpenHevc_Handle h, OpenHevc_FrameInfo *info);
OHEVC_PUBLIC int libOpenHevcGetOutput(OpenHevc_Handle h, int got_picture, OpenHevc_Frame *frame);
OHEVC_PUBLIC int libOpenHevcGetOutputCpy(OpenHevc_Handle h, int got_picture, OpenHevc_Frame_cpy *frame);
OHEVC_PUBLIC void libOpenHevcSetCheckMD5(OpenHevc_Handle h, int val);
OHEVC_PUBLIC void libOpenHevcSetDebugMode(OpenHevc_Handle h, int val);
OHEVC_PUBLIC void libOpenHevcSetTemporalLayer_id(OpenHevc_Handle h, int val);
OHEVC_PUBLIC void libOpenHevcSetNoCropping(OpenHevc_Handle h, int val);
OHEVC_PUBLIC void libOpenHevcSetActiveDecoders(OpenHevc_Handle h, int val);
OHEVC_PUBLIC void libOpenHevcSetViewLayers(OpenHevc_Handle h, int val);
OHEVC_PUBLIC void libOpenHevcClose(OpenHevc_Handle h);
OHEVC_PUBLIC void libOpenHevcFlush(OpenHevc_Handle h);
OHEVC_PUBLIC const char *libOpenHevcVersion(OpenHevc_Handle h);

#ifdef __cplusplus
}
#endif
#endif
