"""audit_load_wait.py [k_file ...]: compiles the device side of the kernel files to ISA text (hipcc -S) and lists, per kernel, how
many vector-memory loads are followed within a few instructions by `s_waitcnt vmcnt(0)` - the sign of a loop of independent loads
that the compiler turned into load -> wait -> store per element (round 3: the tracker's staging walks, 26 dependent round trips
per level and point).  Run from the repo root; needs hipcc only."""
import re, subprocess, sys, os, tempfile
C = "video-stab_amd/csrc"
files = sys.argv[1:] or ["k_lk", "k_gftt", "k_pyr", "k_gray", "k_ransac", "k_warp", "k_traj", "k_roll", "k_azc", "k_enhance", "k_canvas"]
for f in files:
    out = os.path.join(tempfile.gettempdir(), f + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                    "-mllvm", "-amdgpu-kernarg-preload-count=16", "-w", "-S", "--cuda-device-only", "-o", out, os.path.join(C, f + ".hip")], check=True,
                   stderr=subprocess.DEVNULL)
    txt = open(out).read()
    for m in re.finditer(r"^(_ZN[^\n:]+):.*?s_endpgm", txt, re.S | re.M):
        body = m.group(0).split("\n")
        loads = hits = 0
        for i, l in enumerate(body):
            if re.search(r"\b(global|flat)_load", l):
                loads += 1
                for j in range(i + 1, min(i + 8, len(body))):
                    if "s_waitcnt vmcnt(0)" in body[j]:
                        hits += 1
                        break
                    if re.search(r"\b(global|flat)_load", body[j]):
                        break
        if loads:
            name = re.sub(r"^_ZN3vsd12_GLOBAL__N_1\d+", "", m.group(1))
            print("%-12s %-60s loads %3d  load->wait(0) %3d" % (f, name[:60], loads, hits))
