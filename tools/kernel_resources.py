"""Registers, spills, scratch, occupancy and LDS of every kernel from `hipcc -Rpass-analysis=kernel-resource-usage` output
(make -C caps-sa_amd resources 2> file).  usage: kernel_resources.py <file> [--spills] [substring ...]"""
import re, subprocess, sys
args = sys.argv[1:]
only_spills = "--spills" in args
args = [a for a in args if a != "--spills"]
t = open(args[0]).read()
pats = args[1:]
blocks = re.split(r'remark: Function Name: ', t)
names = [b.split()[0] for b in blocks[1:]]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
for b, d in zip(blocks[1:], dem):
    g = lambda k: int((re.search(re.escape(k) + r': (\d+)', b) or [0, 0])[1])   # noqa: E731
    row = {"vgpr": g('VGPRs'), "sgpr": g('TotalSGPRs'), "vgpr_spill": g('VGPRs Spill'), "sgpr_spill": g('SGPRs Spill'),
           "scratch": g('ScratchSize [bytes/lane]'), "occ": g('Occupancy [waves/SIMD]'), "lds": g('LDS Size [bytes/block]')}
    if only_spills and not (row["vgpr_spill"] or row["sgpr_spill"]): continue
    if pats and not any(p in d for p in pats): continue
    d = re.sub(r'\(.*$', '', d.replace('void caps::', '').replace('unsigned int', 'u32').replace('unsigned long', 'u64'))
    print(f"{d[:78]:78s} " + " ".join(f"{k}={v}" for k, v in row.items()))
