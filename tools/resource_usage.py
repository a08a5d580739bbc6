"""Per-kernel register / spill / occupancy table of a hipcc -Rpass-analysis=kernel-resource-usage log (stderr of the compile)."""
import re, subprocess, sys

def parse(path):
    name, d = None, {}
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1); d[name] = {}; continue
        m = re.search(r"remark: .*?\s+(VGPRs|AGPRs|SGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            d[name][m.group(1)] = int(m.group(2))
    return d

if __name__ == "__main__":
    d = parse(sys.argv[1])
    names = subprocess.run(["c++filt"], input="\n".join(d), capture_output=True, text=True).stdout.split("\n")
    for k, dn in zip(d, names):
        v = d[k]
        dn = dn.replace("t3::", "").replace("void ", "")
        print(f"{dn[:100]:100s} V{v.get('VGPRs')} A{v.get('AGPRs')} S{v.get('SGPRs')} spill{v.get('VGPRs Spill')} scratch{v.get('ScratchSize [bytes/lane]')} occ{v.get('Occupancy [waves/SIMD]')}")
