#!/bin/bash
# lab: which part of the direct-W loop disagrees with itself when two systems run on separate streams?
cd ${GRAFT_REPO_ROOT:-/root/repo}
F=physdock_amd/csrc/gemm_split.hip
cp $F /tmp/gs_orig.hip
run() { echo "== $1"; python -m physdock_amd.build --force >/dev/null 2>&1; PD_NO_PRESPLIT_GEMM=1 PD_STEPS=10 python tools/concurrent_ligands.py 20 2>&1 | grep -E "max" | tail -1; }
# (a) B buffers re-requested only after BOTH MFMA groups and the staging VALU (no load right behind its last reader)
python - <<'PY'
p='physdock_amd/csrc/gemm_split.hip'; s=open(p).read()
old='''                mma2(st, 0);
                if (more) wfrag(0, bn0, 2 * kt + 2);          // every B buffer is re-requested right after its last use:
                mma2(st, 1);                                  // a full k-step (+ the staging and the barrier) ahead
                if (more) {
                    wfrag(1, bn0, 2 * kt + 3);
                    // the other stage was last read in the previous iteration, which every wave left through its barrier
                    stage2(st ^ 1, (kt + 1) * 32);'''
new='''                mma2(st, 0);
                mma2(st, 1);
                if (more) {
                    stage2(st ^ 1, (kt + 1) * 32);
                    wfrag(0, bn0, 2 * kt + 2);
                    wfrag(1, bn0, 2 * kt + 3);'''
assert old in s; open(p,'w').write(s.replace(old,new,1))
PY
run "B re-requested after both MFMA groups + staging"
cp /tmp/gs_orig.hip $F
# (b) a full barrier-style wait in front of the re-requests: s_waitcnt on everything + s_nop
python - <<'PY'
p='physdock_amd/csrc/gemm_split.hip'; s=open(p).read()
old='''                if (more) wfrag(0, bn0, 2 * kt + 2);          // every B buffer is re-requested right after its last use:'''
new='''                __builtin_amdgcn_s_sleep(2);
                if (more) wfrag(0, bn0, 2 * kt + 2);          // every B buffer is re-requested right after its last use:'''
assert old in s; s=s.replace(old,new,1)
old='''                    wfrag(1, bn0, 2 * kt + 3);
                    // the other stage'''
new='''                    __builtin_amdgcn_s_sleep(2);
                    wfrag(1, bn0, 2 * kt + 3);
                    // the other stage'''
assert old in s; open(p,'w').write(s.replace(old,new,1))
PY
run "s_sleep(2) (128 cycles) in front of every B re-request"
cp /tmp/gs_orig.hip $F
python -m physdock_amd.build --force >/dev/null 2>&1
