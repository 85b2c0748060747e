#!/bin/bash
# Timing ablations of attention_mfma.hip (results are WRONG by construction; timing only):
#   abl1: no main loops (edges, staging, prologue, epilogue only)   abl2: no edge steps
#   abl3: no staging   abl4: no main loops and no edge steps
set -e
cd "$(dirname "$0")/../.."
SRC=pytorch-generative_amd/csrc/attention_mfma.hip
OBJ=pytorch-generative_amd/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Ipytorch-generative_amd/csrc"
others=$(ls $OBJ/*.o | grep -v attention_mfma.o)
for v in 1 2 3 4; do
  tmp=/tmp/_abl$v.hip
  cp $SRC $tmp
  python3 - $tmp $v <<'PY'
import sys,re
p,v=sys.argv[1],int(sys.argv[2])
s=open(p).read()
if v in (1,4):
    s=s.replace('for (int k0 = 16; k0 < q0; k0 += 16) {','for (int k0 = 16; k0 < 0; k0 += 16) {')
    s=s.replace('for (int k0 = 0; k0 < q0; k0 += 16) {','for (int k0 = 0; k0 < 0; k0 += 16) {')
    s=s.replace('for (int q0t = kb0 + 64; q0t < q_end; q0t += 16) {','for (int q0t = kb0 + 64; q0t < 0; q0t += 16) {')
if v in (2,4):
    lines=s.split('\n')
    for i,l in enumerate(lines):
        if 'step(' in l and 'auto step' not in l and ('B<true>{}' in l or 'I<1>{}, B<false>{}' in l):
            lines[i]=l.replace('step(','noop(')
    s='\n'.join(lines).replace('namespace {\n','#define noop(...) ((void)0)\nnamespace {\n',1)
if v==3:
    s=re.sub(r'for \(int m0 = (0|r0); m0 < (rows|r1); m0 \+= 2 \* blockDim.x\) \{', lambda m: 'for (int m0 = 0; m0 < 0; m0 += 2 * blockDim.x) {', s)
open(p,'w').write(s)
PY
  /opt/rocm/bin/hipcc $FLAGS -c $tmp -o tools/exp/_abl$v.o 2>&1 | grep -E "error" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others tools/exp/_abl$v.o -o tools/exp/libpg_abl$v.so
  rm -f $tmp tools/exp/_abl$v.o
done
ls -la tools/exp/libpg_abl*.so
