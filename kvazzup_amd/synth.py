"""uvgx-synth-v1: integer-only synthetic I420 clips, numpy version.
bench.py feeds these to the encoder; tests/ check them against the C twin in oracle/synth.c."""
import numpy as np

MOVING, FLAT, NOISE = 0, 1, 2


def _fmix32(h):
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def frame(kind, seed, w, h, t):
    """packed I420 frame (uint8, w*h*3/2)"""
    n = w * h * 3 // 2
    seed = np.uint32(seed)
    tk = np.uint32((t * 0x9E3779B1) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        if kind == FLAT:
            return np.full(n, 128, dtype=np.uint8)
        if kind == NOISE:
            i = np.arange(n, dtype=np.uint32)
            return (_fmix32(seed ^ tk ^ (i * np.uint32(0x85EBCA77))) & np.uint32(255)).astype(np.uint8)
        ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
        v = 32 + (xs * 160) // w + (ys * 32) // h
        s = h // 8
        for k in range(8):
            cx = (k * w // 8 + 5 * (k + 1) * t) % w
            cy = (k * h // 8 + 3 * (k + 1) * t) % h
            dx, dy = xs - cx, ys - cy
            m = (dx >= 0) & (dx < s) & (dy >= 0) & (dy < s)
            v = np.where(m, 64 + 16 * k + (((dx * 7) ^ (dy * 13)) & 63), v)
        idx = (ys * w + xs).astype(np.uint32)
        hsh = _fmix32(seed ^ tk ^ (idx * np.uint32(0x85EBCA77)))
        v = v + (hsh & np.uint32(7)).astype(np.int64) - 3
        y = np.clip(v, 16, 235).astype(np.uint8)
        cw, ch = w // 2, h // 2
        cys, cxs = np.mgrid[0:ch, 0:cw].astype(np.int64)
        u = 96 + (cxs * 64) // cw
        vv = 96 + (cys * 64) // ch
        ub = u.copy()
        for k in range(8):
            cx = ((k * w // 8 + 5 * (k + 1) * t) % w) // 2
            cy = ((k * h // 8 + 3 * (k + 1) * t) % h) // 2
            dx, dy = cxs - cx, cys - cy
            m = (dx >= 0) & (dx < s // 2) & (dy >= 0) & (dy < s // 2)
            u = np.where(m, ub + 8 * k, u)
        return np.concatenate([y.reshape(-1), u.astype(np.uint8).reshape(-1), vv.astype(np.uint8).reshape(-1)])


def clip(kind, seed, w, h, frames, start=0):
    return np.stack([frame(kind, seed, w, h, start + t) for t in range(frames)])


def frame_torch(kind, seed, w, h, t, device):
    """same clip generated on the GPU with torch integer ops (bench.py: inputs resident in HBM)"""
    import torch
    M = 0xFFFFFFFF

    def fmix(x):
        x = x ^ (x >> 16)
        x = (x * 0x85EBCA6B) & M
        x = x ^ (x >> 13)
        x = (x * 0xC2B2AE35) & M
        x = x ^ (x >> 16)
        return x

    n = w * h * 3 // 2
    tk = (t * 0x9E3779B1) & M
    if kind == FLAT:
        return torch.full((n,), 128, dtype=torch.uint8, device=device)
    if kind == NOISE:
        i = torch.arange(n, dtype=torch.int64, device=device)
        return (fmix((seed ^ tk) ^ ((i * 0x85EBCA77) & M)) & 255).to(torch.uint8)
    ys = torch.arange(h, dtype=torch.int64, device=device).view(h, 1)
    xs = torch.arange(w, dtype=torch.int64, device=device).view(1, w)
    v = (32 + (xs * 160) // w + (ys * 32) // h).expand(h, w).clone()
    s = h // 8
    for k in range(8):
        cx = (k * w // 8 + 5 * (k + 1) * t) % w
        cy = (k * h // 8 + 3 * (k + 1) * t) % h
        dx, dy = xs - cx, ys - cy
        m = (dx >= 0) & (dx < s) & (dy >= 0) & (dy < s)
        v = torch.where(m, 64 + 16 * k + (((dx * 7) ^ (dy * 13)) & 63), v)
    idx = ys * w + xs
    hsh = fmix((seed ^ tk) ^ ((idx * 0x85EBCA77) & M))
    v = v + (hsh & 7) - 3
    y = v.clamp(16, 235).to(torch.uint8)
    cw, ch = w // 2, h // 2
    cys = torch.arange(ch, dtype=torch.int64, device=device).view(ch, 1)
    cxs = torch.arange(cw, dtype=torch.int64, device=device).view(1, cw)
    ub = (96 + (cxs * 64) // cw).expand(ch, cw)
    vv = (96 + (cys * 64) // ch).expand(ch, cw)
    u = ub.clone()
    for k in range(8):
        cx = ((k * w // 8 + 5 * (k + 1) * t) % w) // 2
        cy = ((k * h // 8 + 3 * (k + 1) * t) % h) // 2
        dx, dy = cxs - cx, cys - cy
        m = (dx >= 0) & (dx < s // 2) & (dy >= 0) & (dy < s // 2)
        u = torch.where(m, ub + 8 * k, u)
    return torch.cat([y.reshape(-1), u.to(torch.uint8).reshape(-1), vv.to(torch.uint8).reshape(-1)])


def scene_cut_frame(seed, w, h, t, cut):
    """the moving-objects clip with a scene cut at picture `cut`: from there on the picture is the clip's mirror image (left-right and
    up-down) 400 pictures later -- nothing of it is in the reference picture.  Used for the intra-in-P tests and numbers."""
    if t < cut:
        return frame(MOVING, seed, w, h, t)
    f = frame(MOVING, seed ^ 0x00C0FFEE, w, h, t + 400)
    y = f[:w * h].reshape(h, w)[::-1, ::-1]
    u = f[w * h:w * h + w * h // 4].reshape(h // 2, w // 2)[::-1, ::-1]
    v = f[w * h + w * h // 4:].reshape(h // 2, w // 2)[::-1, ::-1]
    return np.concatenate([y.reshape(-1), u.reshape(-1), v.reshape(-1)]).astype(np.uint8)
