"""Per-kernel register / LDS / scratch usage of a HIP source, from hipcc's own remarks (-Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py kbot-joystick_amd/csrc/kbj_nn.hip [name filter ...] [-- extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
src, filters = args[0], args[1:]
cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{ROOT}/include", f"-I{os.path.dirname(os.path.abspath(src))}", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + extra
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
names = re.findall(r"Function Name: (\S+)", txt)
dem = dict(zip(names, subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines())) if names else {}
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].split()[0].strip()
    d = dem.get(name, name)
    if filters and not all(f in d for f in filters):
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    sc, lds, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{d[:100]:100s} VGPR {g('VGPRs'):>3s} AGPR {g('AGPRs'):>3s} spill {g('VGPR Spill'):>3s} scratch {sc:>4s} LDS {lds:>6s} occ {occ}")
