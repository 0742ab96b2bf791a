"""Register / scratch / LDS / occupancy table of the kernels in one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kres.py eae_conv_launch [name-filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hybrid-autoencoder-mlp-pipeline-for-satellite-image-classification_amd", "csrc")


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src + ".hip", "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:], cwd=CSRC, capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name"):
            name = t.split(": ")[1]
            d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            cur = {"name": d}
        for k in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
            if t.startswith(k + ":"):
                cur[k] = t.split(": ")[1]
        if t.startswith("LDS Size") and flt in cur["name"]:
            print(f"{cur['name'][:84]:84s} vgpr {cur.get('VGPRs'):>3s} agpr {cur.get('AGPRs'):>3s} scratch {cur.get('ScratchSize [bytes/lane]'):>4s} "
                  f"occ {cur.get('Occupancy [waves/SIMD]')} lds {cur.get('LDS Size [bytes/block]')}")


if __name__ == "__main__":
    main()
