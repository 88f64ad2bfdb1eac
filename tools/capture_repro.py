"""Diagnostic for the capture_end crash of round 2 (gpurun_out/r2_gs.log): plain PyTorch only, nothing of this library.

Hypothesis: a leaf's AccumulateGrad node is created lazily, under the stream that is current when the
leaf first enters an autograd graph, and stays alive as long as ANY autograd graph that reaches the leaf does.  After
eager steps on the default stream whose outputs are still referenced, a backward captured on a side stream hands its
gradients to that old node: the engine makes the node's stream — the legacy default stream — wait on an event recorded
in the capturing stream, which pulls the default stream into the capture.  Each case below runs in its own process
(a crash must not take the caller down) and prints its exit code:

  stale   eager graph alive, capture on a side stream            <- the crash of round 2, if the hypothesis holds
  fresh   eager graph released first (what round 2's `del` did)
  alias   eager graph alive, capture through fresh aliases of the leaves (cuda_kernel.GraphedStep does this)

  python tools/capture_repro.py            # runs all cases, one subprocess each
"""
import subprocess
import sys
import warnings


def case(name):
    import torch

    dev = torch.device("cuda", 0)
    leaf = torch.randn(4096, device=dev, requires_grad=True)
    eager = (leaf * 2.0).sum()  # noqa: F841  an autograd graph of an earlier eager step, still referenced
    if name == "fresh":
        del eager
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())

    def step():
        src = leaf.detach().requires_grad_(True) if name == "alias" else leaf
        return torch.autograd.grad((src * 3.0).sum(), [src])[0]

    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            g = step()
        with torch.no_grad():
            leaf.mul_(2.0)
        graph.replay()
        torch.cuda.synchronize()
    ok = bool(torch.equal(g, torch.full_like(g, 3.0)))
    mism = [str(w.message)[:60] for w in rec if "AccumulateGrad" in str(w.message)]
    print(f"case {name}: captured and replayed, gradient correct = {ok}, AccumulateGrad warnings = {len(mism)}", flush=True)


def main():
    if len(sys.argv) > 1:
        case(sys.argv[1])
        return
    for name in ("fresh", "alias", "stale"):
        res = subprocess.run([sys.executable, "-X", "faulthandler", __file__, name], capture_output=True, text=True, timeout=300)
        tail = (res.stdout + res.stderr).strip().splitlines()
        keep = [ln for ln in tail if ln.startswith("case ") or "Error" in ln or "Fatal" in ln or "capture_end" in ln or "hip" in ln.lower()][:8]
        print(f"== {name}: exit code {res.returncode}")
        for ln in keep:
            print("   ", ln)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
