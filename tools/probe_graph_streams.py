"""Probe of the hipGraph replay crash (eabnet_amd/graphs.py).

hip::Graph::UpdateStreams skips every parallel stream of a graph that maps to the same device queue as the launch stream
and does not bound the index it advances.  If that reading of the disassembly is right, a graph with internal branches
must crash deterministically once ALL streams of the process share one hardware queue (GPU_MAX_HW_QUEUES=1), and must not
with single-stream graphs.  Children are separate processes (a segmentation fault ends only the child); the kernels are
trivial element-wise torch ops.

    python tools/probe_graph_streams.py            -> one line per (queues, branches) with the child's exit status
"""
import os
import subprocess
import sys

CHILD = r"""
import sys, torch
branches = int(sys.argv[1])
dev = torch.device("cuda:0")
x = torch.zeros(1 << 16, device=dev)
side = [torch.cuda.Stream(device=dev) for _ in range(branches)]
outs = [torch.zeros_like(x) for _ in range(branches + 1)]
def body():
    main = torch.cuda.current_stream()
    outs[0].copy_(x).add_(1.0)
    for s in side:
        s.wait_stream(main)
    for k, s in enumerate(side):
        with torch.cuda.stream(s):
            outs[k + 1].copy_(x).mul_(2.0).add_(float(k))
    for s in side:
        main.wait_stream(s)
    outs[0].add_(sum(outs[1:]))
body(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
for extra in range(int(sys.argv[2])):          # streams created after the capture (other libraries do that)
    torch.cuda.Stream(device=dev)
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
print("ok", float(outs[0][0]))
"""


def main() -> None:
    for queues in ("default", "4", "2", "1"):
        for branches in (0, 2, 3):
            env = dict(os.environ)
            if queues != "default":
                env["GPU_MAX_HW_QUEUES"] = queues
            r = subprocess.run([sys.executable, "-c", CHILD, str(branches), "0"], env=env, capture_output=True, text=True, timeout=300)
            tail = (r.stdout.strip().splitlines() or [""])[-1]
            err = [l for l in r.stderr.splitlines() if "Segmentation" in l or "Error" in l][:1]
            print(f"GPU_MAX_HW_QUEUES={queues:8s} branches={branches}: exit {r.returncode:4d}  {tail} {err}", flush=True)


if __name__ == "__main__":
    main()
