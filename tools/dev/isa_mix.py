"""Developer script (no GPU needed): instruction mix per basic block of one kernel in the gfx950 ISA of kernels.hip.
usage: python tools/dev/isa_mix.py <mangled-name-substring> [min MFMAs per block to print]
(compiles iwae_amd/csrc/kernels.hip to /tmp/iwae_k.s once per call)"""
import collections, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                       "-o", "/tmp/iwae_k.s", os.path.join(root, "iwae_amd", "csrc", "kernels.hip")], stderr=subprocess.DEVNULL)
s = open("/tmp/iwae_k.s").read()
pat, minm = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", s, re.S | re.M):
    if pat not in m.group(1):
        continue
    print("==", m.group(1))
    parts = re.split(r"\n(\.LBB\d+_\d+):", m.group(2))
    blocks = [("entry", parts[0])] + [(parts[i], parts[i + 1]) for i in range(1, len(parts), 2)]
    for lab, b in blocks:
        ins = [l.strip().split()[0] for l in b.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
        c = collections.Counter()
        for x in ins:
            c["mfma" if x.startswith("v_mfma") else "valu" if x.startswith("v_") else "salu" if x.startswith("s_") else
              "lds" if x.startswith("ds_") else "vmem" if x.startswith(("global_", "buffer_", "scratch_")) else "other"] += 1
        if c["mfma"] >= minm:
            print(lab, dict(c))
            print("    ", collections.Counter(x for x in ins if x.startswith("v_") and not x.startswith("v_mfma")).most_common(16))
