"""Copy the round-2 measurement set (scratch/collect_r02.sh -> gpurun_out/r02/) into profiles/r02_*, keeping the header
comments ('#' lines) of the files already there and replacing their data lines."""
import glob, os, shutil, sys
src = "gpurun_out/r02"
def header(path):
    if not os.path.exists(path): return []
    return [l for l in open(path).read().splitlines() if l.startswith("#")]
for f in sorted(glob.glob(os.path.join(src, "*"))):
    name = os.path.basename(f)
    if name in ("prof_r02.log", "prof_evi.log") or os.path.getsize(f) == 0: continue
    dst = os.path.join("profiles", "r02_" + name)
    body = [l for l in open(f).read().splitlines() if not l.startswith("/opt/amdgpu")]
    if name.endswith(".txt"):
        hdr = header(dst)
        open(dst, "w").write("\n".join(hdr + [l for l in body if not l.startswith("#") or l not in hdr]) + "\n")
    else:
        open(dst, "w").write("\n".join(body) + "\n")
    print("profiles/r02_" + name)
