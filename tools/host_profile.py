"""Where the HOST time of one training step goes: cProfile over a few steps enqueued without any fence
in between, sorted by own time.  Run on the GPU box:  python tools/host_profile.py [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device('cuda', 0)
    model = bench.build_model(dev)
    from packnet_sfm.rccl.flat_adam import FlatAdam
    opt = FlatAdam([{'name': 'Depth', 'params': list(model.depth_net.parameters()), 'lr': 2e-4},
                    {'name': 'Pose', 'params': list(model.pose_net.parameters()), 'lr': 2e-4}])
    batch = bench.synthetic_batch(4, 192, 640, 1234, dev)

    def step():
        opt.zero_grad()
        out = model(batch, progress=0.0)
        out['loss'].backward()
        opt.step()

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    # plain wall time of the enqueue (empty queue in front)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    print('host enqueue per step (no profiler): %s ms' % ['%.2f' % (1e3 * t) for t in ts])
    # forward-only / backward-only split
    torch.cuda.synchronize()
    t0 = time.perf_counter(); opt.zero_grad(); out = model(batch, progress=0.0); t1 = time.perf_counter()
    out['loss'].backward(); t2 = time.perf_counter(); opt.step(); t3 = time.perf_counter()
    print('forward %.2f ms, backward %.2f ms, optimizer %.2f ms' % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
    torch.cuda.synchronize()
    try:    # run backward on the calling thread so that cProfile sees the Python of the backward Functions
        torch.autograd.set_multithreading_enabled(False)
    except Exception as e:
        print('set_multithreading_enabled failed:', e)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        torch.cuda.synchronize()
        step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s)
    st.sort_stats('tottime').print_stats(70)
    print(s.getvalue()[:14000])
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(60)
    print(s.getvalue()[:12000])


if __name__ == '__main__':
    main()
