"""Per-source-line histogram of tools/cpu_sampler.c samples that fall into one library (built with -g).
   python tools/cpu_sampler_report.py gpurun_out/cpu_samples.txt kvazzup_amd/libkvazzup_amd_g.so [top]"""
import collections, subprocess, sys
path, lib = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
only_tid = int(sys.argv[4]) if len(sys.argv) > 4 else None      # samples of one thread only
base = None
name = lib.split("/")[-1]
maps, samples, callers = [], [], []
for line in open(path):
    if line.startswith("M "):
        f = line.split()
        lo, hi = (int(x, 16) for x in f[1].split("-"))
        maps.append((lo, hi, int(f[3], 16), f[-1]))
    elif line.startswith("S "):
        f = line.split()
        if only_tid is None or int(f[2]) == only_tid:
            samples.append(int(f[1], 16))
            callers.append(int(f[3], 16) if len(f) > 3 else 0)
# file offset -> virtual address of the executable segment (what the symbolizer wants)
delta = 0
for line in subprocess.run(["readelf", "-lW", lib], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout.splitlines():
    f = line.split()
    if f and f[0] == "LOAD" and "E" in "".join(f[6:8]):
        delta = int(f[2], 16) - int(f[1], 16)
inlib = collections.Counter()
other = collections.Counter()
outside = collections.Counter()          # (library the sample fell into, nearest exported symbol there, return address into the measured library found on the stack)
def lib_va(a):
    for lo, hi, off, nm in maps:
        if lo <= a < hi and nm.endswith(name):
            return a - lo + off + delta
    return None
_dyn = {}
def nearest(path, fileoff_va):
    import bisect
    if path not in _dyn:
        tab = []
        for l in subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout.splitlines():
            f = l.split()
            if len(f) >= 3 and f[1] in "TtWwi":
                tab.append((int(f[0], 16), f[2].split("@")[0]))
        tab.sort(); _dyn[path] = tab
    tab = _dyn[path]
    i = bisect.bisect_right([a for a, _ in tab], fileoff_va) - 1
    return tab[i][1] if i >= 0 else "?"
bases = {}
for lo, hi, off, nm in maps:
    bases[nm] = min(bases.get(nm, lo - off), lo - off)
for pc, ca in zip(samples, callers):
    for lo, hi, off, nm in maps:
        if lo <= pc < hi:
            if nm.endswith(name):
                inlib[pc - lo + off + delta] += 1
            else:
                other[nm.split("/")[-1]] += 1
                import os
                outside[(nm.split("/")[-1], nearest(nm, pc - bases[nm]) if os.path.exists(nm) else "?", lib_va(ca) if ca else None)] += 1
            break
    else:
        other["?"] += 1
print("samples %d; in %s: %d" % (len(samples), name, sum(inlib.values())))
for nm, n in other.most_common(12):
    print("  %6d  %s" % (n, nm))
addrs = list(inlib)
SYM = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"
out = subprocess.run([SYM, "--obj=" + lib, "-f", "-C", "-i", "-a"] + [hex(a) for a in addrs], capture_output=True, text=True).stdout.splitlines()
by_line, by_func, by_outer = collections.Counter(), collections.Counter(), collections.Counter()
cur, frames = None, []
def flush():
    if cur is None or not frames:
        return
    n = inlib[cur]
    fn, loc = frames[0]
    by_line[(fn, ":".join(loc.split("/")[-1].split(":")[:2]))] += n
    by_func[fn] += n
    by_outer[frames[-1][0]] += n
i = 0
while i < len(out):
    if out[i].startswith("0x"):
        flush(); cur = int(out[i], 16); frames = []; i += 1
    elif not out[i].strip():
        i += 1
    else:
        frames.append((out[i], out[i + 1] if i + 1 < len(out) else "?")); i += 2
flush()
tot = sum(inlib.values())
print("-- by outermost function")
for k, n in by_outer.most_common(15):
    print("  %5.1f%%  %s" % (100.0 * n / tot, k[:140]))
print("-- by innermost (inlined) function")
for k, n in by_func.most_common(top):
    print("  %5.1f%%  %s" % (100.0 * n / tot, k[:140]))
print("-- by line")
for (fn, loc), n in by_line.most_common(top):
    print("  %5.1f%%  %-28s %s" % (100.0 * n / tot, loc, fn[:100]))

# samples outside the library: which of its functions was (most likely) waiting for the call to return
if outside:
    cas = sorted({c for (_, _, c) in outside if c})
    names = {}
    if cas:
        o2 = subprocess.run([SYM, "--obj=" + lib, "-f", "-C", "-a"] + [hex(a) for a in cas], capture_output=True, text=True).stdout.splitlines()
        j = 0
        while j < len(o2):
            if o2[j].startswith("0x"):
                names[int(o2[j], 16)] = (o2[j + 1] if j + 1 < len(o2) else "?", o2[j + 2].split("/")[-1] if j + 2 < len(o2) else "?"); j += 3
            else:
                j += 1
    agg = collections.Counter()
    for (l, sym, c), n in outside.items():
        fn, loc = names.get(c, ("(no frame of the library on the stack)", ""))
        agg[(l, sym, fn[:90], ":".join(loc.split(":")[:2]))] += n
    print("-- outside the library: library, nearest exported symbol, innermost frame of the library on the stack (stack scan, not an unwind)")
    alls = len(samples)
    for (l, sym, fn, loc), n in agg.most_common(40):
        print("  %5.1f%%  %-22s %-28s <- %s %s" % (100.0 * n / alls, l[:22], sym[:28], fn, loc))
