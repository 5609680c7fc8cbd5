"""What one DEPENDENT launch costs inside a replayed hipGraph (diagnostics): K kernels in a chain on one stream, captured once, replayed.
  * tiny kernels (add on 64 floats): time per node = the dependent-launch latency itself;
  * streaming kernels (y = x + 1 over M MB, out of place, ping-pong): time per node against the same kernel's back-to-back rate in a chain of
    independent launches is not measurable on one stream, so the number to read is (time per node) - (bytes / the rate of a long kernel);
  * the same chains on two forked streams inside one graph: do the boundaries of one chain hide behind the other chain's kernels?
usage: python tools/chain_probe.py > profiles/r05_chain_probe.txt"""
import torch

dev = torch.device("cuda:0")
K = 200


def replay_time(build, reps=20):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        build()          # warm-up (allocations, kernel selection)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            build()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3      # us per replay


def chain(bufs, k):
    for i in range(k):
        torch.add(bufs[i & 1], 1.0, out=bufs[(i + 1) & 1])


def two_chains(b0, b1, k):
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        chain(b1, k)
    chain(b0, k)
    cur.wait_stream(side)


def main_plus_stub(b0, b1, k, stub):
    """a chain of k nodes beside a side branch of `stub` nodes forked at the start and joined at the end"""
    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        chain(b1, stub)
    chain(b0, k)
    cur.wait_stream(side)


print(f"# tools/chain_probe.py: {K} dependent kernels per chain in one captured graph, us per NODE of a chain (one MI355X)")
print("# MB per kernel (in + out) | one chain | two chains on forked streams (per node of ONE chain: equal to the first column = the second chain is free)")
for mb in ((0.0005, 13) if __import__("os").environ.get("CHAIN_FAST") else (0.0005, 1, 4, 13, 26, 51, 103)):
    n = max(64, int(mb * 1e6 / 8))
    b0 = [torch.zeros(n, device=dev), torch.zeros(n, device=dev)]
    b1 = [torch.zeros(n, device=dev), torch.zeros(n, device=dev)]
    t1 = replay_time(lambda: chain(b0, K)) / K
    t2 = replay_time(lambda: two_chains(b0, b1, K)) / K
    t3 = replay_time(lambda: main_plus_stub(b0, b1, K, 1)) / K
    t4 = replay_time(lambda: main_plus_stub(b0, b1, K, 20)) / K
    print(f"{mb:9.4f} MB   {t1:7.2f} us   {t2:7.2f} us   | chain + a 1-node side branch {t3:7.2f} us, + a 20-node side branch {t4:7.2f} us", flush=True)
