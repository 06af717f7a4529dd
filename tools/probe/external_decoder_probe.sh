#!/bin/bash
# tools/probe/external_decoder_probe.sh -- lists everything on the GPU box that could decode HEVC and was not written here
# (VERDICT r5 next #1).  Plain shell, queries only.  Output: gpurun_out/external_decoder_probe.txt
out=gpurun_out/external_decoder_probe.txt; mkdir -p gpurun_out
{
echo "== date/host"; date -u; uname -r; id
echo "== device nodes"; ls -la /dev/dri /dev/kfd 2>&1
echo "== libraries"
ls -la /opt/rocm*/lib/librocdecode* /opt/rocm*/lib/librocjpeg* /opt/rocm*/lib/libamf* 2>&1
ls -la /usr/lib/x86_64-linux-gnu/libva* /usr/lib/x86_64-linux-gnu/dri/*_drv_video.so /usr/lib/x86_64-linux-gnu/libvdpau* /usr/lib/x86_64-linux-gnu/vdpau 2>&1
ls -la /usr/lib/x86_64-linux-gnu/libavcodec* /usr/lib/x86_64-linux-gnu/libde265* /usr/lib/x86_64-linux-gnu/libx265* /usr/lib/x86_64-linux-gnu/libkvazaar* /usr/lib/x86_64-linux-gnu/libgst* /usr/lib/x86_64-linux-gnu/libheif* 2>&1
echo "== find (whole filesystem, names)"
find / -xdev \( -iname '*rocdecode*.so*' -o -iname '*libva.so*' -o -iname '*drv_video*' -o -iname '*avcodec*' -o -iname '*de265*' -o -iname '*x265*' -o -iname '*kvazaar*' -o -iname '*openhevc*' -o -iname '*hevc*' -o -iname '*h265*' -o -iname '*libheif*' -o -iname '*gstlibav*' -o -iname '*libamf*' -o -iname '*vcn*' \) -not -path '/proc/*' -not -path '/sys/*' -not -path "$PWD/*" -not -path '/root/repo/*' 2>/dev/null | head -80
echo "== firmware"; ls /lib/firmware/amdgpu 2>&1 | grep -i -E 'vcn|uvd' | head
echo "== programs"; for p in ffmpeg ffprobe gst-launch-1.0 gst-inspect-1.0 vainfo vdpauinfo x265 kvazaar mpv vlc mplayer HandBrakeCLI; do printf '%s: ' $p; command -v $p || echo absent; done
echo "== python modules"
python3 - <<'PY'
import importlib
for m in ['av', 'cv2', 'torchvision', 'torchvision.io', 'torchcodec', 'decord', 'imageio', 'imageio_ffmpeg', 'skvideo', 'pyrocdecode', 'rocpydecode', 'pillow_heif', 'gi']:
    try:
        importlib.import_module(m); print(m, 'present')
    except Exception as e:
        print(m, 'absent', type(e).__name__)
try:
    import gi
    for ns in ['Gst', 'GstVideo']:
        try: gi.require_version(ns, '1.0'); print('gi', ns, 'present')
        except Exception as e: print('gi', ns, 'absent')
except Exception: pass
PY
echo "== torch build: video support?"
python3 - <<'PY'
import torch
print(torch.__version__, [n for n in dir(torch.ops) if 'video' in n.lower() or 'decode' in n.lower()])
PY
echo "== amdgpu video IP blocks (kernel query, no submission)"
./tools/probe/probe_vcn 2>&1
echo "== rocminfo (agents)"; /opt/rocm/bin/rocminfo 2>&1 | grep -E 'Marketing Name|Name:.*gfx|Compute Unit' | head
echo "== sysfs"; for d in /sys/class/drm/card*/device; do echo $d; cat $d/vendor $d/device 2>/dev/null; ls $d 2>/dev/null | grep -i -E 'vcn|uvd|jpeg|ip_discovery' ; ls $d/ip_discovery/die/0 2>/dev/null | tr '\n' ' '; echo; done 2>&1 | head -60
} > $out 2>&1
tail -n 60 $out
