O=gpurun_out/r6as; mkdir -p $O
export TMPDIR=/tmp
export SSX_BENCH_SINGLE_GPU_GLOO=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo "rc=$?"
tail -c 1500 $O/bench_2ranks.json; echo; tail -5 $O/bench_2ranks.err | cut -c1-300
timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --lean > $O/bench_2ranks_self.json 2> $O/bench_2ranks_self.err; echo "self-launch rc=$?"
python - <<'P'
import json
for f in ("gpurun_out/r6as/bench_2ranks.json", "gpurun_out/r6as/bench_2ranks_self.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["n_gpus"], d["value"], d.get("nccl_ranks"), d["ba_c4"].get("collective"), d["ba_c4"].get("sharding"), d["ba_c4"]["weak"]["iters_per_s"])
    except Exception as e:
        print(f, "FAILED", e)
P
