/* tools/probe/probe_vcn.c -- asks the amdgpu kernel driver which video IP blocks the device exposes (query only: no command submission).
 * Part of the round-6 search for an HEVC decoder nobody here wrote.  Build: gcc -O1 probe_vcn.c -ldrm_amdgpu -ldrm */
#include <stdio.h>
#include <string.h>
#include <fcntl.h>
#include <unistd.h>
#include <amdgpu.h>
#include <amdgpu_drm.h>
int main(void)
{
    int found = 0;
    for (int n = 128; n < 256; n++) {
        char path[64];
        snprintf(path, sizeof path, "/dev/dri/renderD%d", n);
        int fd = open(path, O_RDWR);
        if (fd < 0) { if (access(path, F_OK) == 0) printf("%s: exists, open failed (permissions)\n", path); continue; }
        uint32_t maj, min;
        amdgpu_device_handle dev;
        if (amdgpu_device_initialize(fd, &maj, &min, &dev)) { printf("%s: not an amdgpu device\n", path); close(fd); continue; }
        found++;
        struct amdgpu_gpu_info gi; memset(&gi, 0, sizeof gi);
        amdgpu_query_gpu_info(dev, &gi);
        printf("%s: amdgpu drm %u.%u family %u chip_rev %u\n", path, maj, min, gi.family_id, gi.chip_rev);
        static const struct { unsigned ip; const char* name; } ips[] = {
            {AMDGPU_HW_IP_UVD, "UVD"}, {AMDGPU_HW_IP_VCE, "VCE"}, {AMDGPU_HW_IP_UVD_ENC, "UVD_ENC"},
            {AMDGPU_HW_IP_VCN_DEC, "VCN_DEC"}, {AMDGPU_HW_IP_VCN_ENC, "VCN_ENC(unified)"}, {AMDGPU_HW_IP_VCN_JPEG, "VCN_JPEG"}};
        for (unsigned i = 0; i < sizeof ips / sizeof ips[0]; i++) {
            struct drm_amdgpu_info_hw_ip info; memset(&info, 0, sizeof info);
            unsigned count = 0;
            int r = amdgpu_query_hw_ip_info(dev, ips[i].ip, 0, &info);
            amdgpu_query_hw_ip_count(dev, ips[i].ip, &count);
            printf("  %-18s rc %d instances %u rings 0x%x ip version %u.%u\n", ips[i].name, r, count, info.available_rings, info.hw_ip_version_major, info.hw_ip_version_minor);
        }
        amdgpu_device_deinitialize(dev);
        close(fd);
    }
    if (!found) printf("no amdgpu render node could be opened\n");
    return 0;
}
