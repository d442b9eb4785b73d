"""Register / scratch / occupancy table of every kernel in libxicsrt_hip.so (development aid, no GPU needed):
    XRT_EXTRA_FLAGS=-Rpass-analysis=kernel-resource-usage bash xicsrt_amd/csrc/build.sh > /tmp/res.log 2>&1
    python tests/tools/resource_usage.py /tmp/res.log"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
for b in blocks:
    name = b.split('\n')[0].split()[0]
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    try:
        name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    name = name.replace('void ', '')
    name = re.sub(r'\(.*', '', name)[:60]
    print('%-60s vgpr %3d agpr %3d scratch %4d sgpr %3d vspill %3d sspill %3d waves/simd %d lds %6d' % (
        name, g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'), g('SGPRs'), g('VGPRs Spill'), g('SGPRs Spill'),
        g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
