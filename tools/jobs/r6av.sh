O=gpurun_out/r6av; mkdir -p $O
export TMPDIR=/tmp
bash tools/ab_variants.sh new ilp memcl bias0 norp 2>&1 | grep -E "per batch|k_lin_schur|k_backsub|k_solve" | tee $O/sched_ab.txt
python tools/solve64_ab.py /tmp/new.npz > /dev/null 2>&1
for v in ilp memcl bias0 norp; do SSX_LIB=$PWD/ssvio_amd/libssx.so.$v python tools/solve64_ab.py /tmp/$v.npz > /dev/null 2>&1; echo -n "$v vs new: "; python tools/solve64_ab.py --compare /tmp/new.npz /tmp/$v.npz; done | tee -a $O/sched_ab.txt
