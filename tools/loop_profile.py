"""Static profile of the step kernel's GJK loop: VALU instructions of the largest loop of env_kernel<Dyn, STEP>, attributed to
source functions through the line table (innermost inlined location).  Diagnostic; usage:
    hipcc <flags of the Makefile> -gline-tables-only -S --cuda-device-only -o build/prof/dev.s urgym_hip.hip
    python tools/loop_profile.py ur_gym_amd/csrc/build/prof/dev.s"""
import re, collections, bisect, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_ZN12_GLOBAL__N_110env_kernelILi2ELi0ELb0EEEvNS_7KParamsEPKf:')][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')][0]
body = lines[start:end]
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r's_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i: loops.append((labels[t], i))
a, b = max(loops, key=lambda x: x[1] - x[0])
dev = open(os.path.join(ROOT, 'ur_gym_amd/csrc/urgym_device.h')).read().split('\n')
fn = [(i, m.group(1)) for i, l in enumerate(dev, 1) for m in [re.match(r'^__device__ __forceinline__ .*?(\w+)\(', l)] if m]
def fname(f, ln):
    if f == 'urgym_device.h':
        k = bisect.bisect_right([x[0] for x in fn], ln) - 1
        return 'dev:' + (fn[k][1] if k >= 0 else '?')
    if f == 'urgym_hip.hip': return 'hip:%d' % (ln // 20 * 20)
    return f + ':%d' % ln
cur = None; valu = collections.Counter(); allc = collections.Counter()
for i, l in enumerate(body):
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (files.get(int(m.group(1)), '?'), int(m.group(2))); continue
    if i < a or i > b: continue
    t = l.strip()
    if not t or t.startswith('.') or t.startswith(';') or t.endswith(':'): continue
    k = fname(*cur) if cur else '?'
    allc[k] += 1
    if t.startswith('v_'): valu[k] += 1
print("loop VALU", sum(valu.values()), "all", sum(allc.values()))
for k, c in valu.most_common(30): print("%-40s valu %5d  all %5d" % (k, c, allc[k]))
