"""Batches in flight (ndtpso_set_pipeline_depth / ndtpso_pipeline_flush): consecutive ndtpso_align_pairs_dev calls of one
context overlap on the device and return exactly what they return one at a time."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _buffers(torch, dev, p, B):
    d = {}
    d["ref"] = torch.from_numpy(p.ref_ranges).to(dev)
    d["new"] = torch.from_numpy(p.new_ranges).to(dev)
    d["guess"] = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    d["dev"] = torch.tensor((0.1, 0.1, 3.1415e-3), dtype=torch.float64, device=dev).repeat(B, 1).contiguous()
    d["seeds"] = torch.from_numpy(p.seeds.astype(np.int64)).to(dev).to(torch.int32)
    return d


def _outs(torch, dev, B):
    return (torch.zeros(B, 3, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.float64, device=dev),
            torch.zeros(B, 8, dtype=torch.int32, device=dev))


@pytest.mark.parametrize("mode_name", ["exact", "f32", "f64"])
def test_two_batches_in_flight_equal_one_at_a_time(mode_name):
    import torch
    from ndtpso_slam_amd import capi, synth
    dev = torch.device("cuda", 0)
    mode = {"exact": capi.SCORE_EXACT, "f32": capi.SCORE_F32, "f64": capi.SCORE_F64}[mode_name]
    B, P, I = 160, 30, 20     # more pairs than half the compute units: the pipelined path, not the cluster path
    batches = [synth.make_pairs(B, seed=40 + k) for k in range(5)]
    geom = capi.ScanGeom(batches[0].n_beams, float(batches[0].angle_min), float(batches[0].angle_inc), float(batches[0].range_max), 0.1)
    grid, cfg = capi.Grid(60, 60, 0.5), capi.PSOConfig.make(I, P)
    ctx = capi.Context(0)
    stream = torch.cuda.Stream(dev)
    ctx.set_stream(stream.cuda_stream)
    ins = [_buffers(torch, dev, p, B) for p in batches]
    torch.cuda.synchronize()

    def run(depth):
        ctx.set_pipeline_depth(depth)
        outs = [_outs(torch, dev, B) for _ in batches]
        for d, (po, co, st) in zip(ins, outs):
            ctx.align_pairs_dev(B, d["ref"].data_ptr(), d["new"].data_ptr(), geom, grid, d["guess"].data_ptr(), d["dev"].data_ptr(),
                                cfg, d["seeds"].data_ptr(), 0, mode, po.data_ptr(), co.data_ptr(), st.data_ptr())
        ctx.synchronize()      # flushes the lanes, then waits for the context's stream
        return [(po.cpu().numpy(), co.cpu().numpy(), st.cpu().numpy().view(capi.STATS_DTYPE).reshape(B)) for po, co, st in outs]

    serial = run(1)
    piped = run(2)
    for (p1, c1, s1), (p2, c2, s2) in zip(serial, piped):
        assert np.array_equal(p1, p2) and np.array_equal(c1, c2)
        assert np.array_equal(s1["status"], s2["status"]) and np.array_equal(s1["gbest_updates"], s2["gbest_updates"])
        assert (s2["status"] & 0xffff == 0).all()
    ctx.set_pipeline_depth(1)
    ctx.close()


def test_flush_orders_outputs_on_the_context_stream():
    """pipeline_flush(keep_newest=1) makes the context's stream wait for every call but the newest: a device-side copy
    of call k - 1's poses enqueued on the context's stream right after launching call k must see the finished poses."""
    import torch
    from ndtpso_slam_amd import capi, synth
    dev = torch.device("cuda", 0)
    B, P, I = 200, 30, 30
    p = synth.make_pairs(B, seed=77)
    geom = capi.ScanGeom(p.n_beams, float(p.angle_min), float(p.angle_inc), float(p.range_max), 0.1)
    grid, cfg = capi.Grid(60, 60, 0.5), capi.PSOConfig.make(I, P)
    ctx = capi.Context(0)
    stream = torch.cuda.Stream(dev)
    ctx.set_stream(stream.cuda_stream)
    d = _buffers(torch, dev, p, B)
    want, _, _ = _outs(torch, dev, B)
    st0 = torch.zeros(B, 8, dtype=torch.int32, device=dev)
    co0 = torch.zeros(B, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    ctx.align_pairs_dev(B, d["ref"].data_ptr(), d["new"].data_ptr(), geom, grid, d["guess"].data_ptr(), d["dev"].data_ptr(), cfg,
                        d["seeds"].data_ptr(), 0, capi.SCORE_EXACT, want.data_ptr(), co0.data_ptr(), st0.data_ptr())
    ctx.synchronize()
    ctx.set_pipeline_depth(2)
    steps = 6
    outs = [_outs(torch, dev, B) for _ in range(2)]
    copies = []
    with torch.cuda.stream(stream):
        for k in range(steps):
            po, co, st = outs[k & 1]
            ctx.align_pairs_dev(B, d["ref"].data_ptr(), d["new"].data_ptr(), geom, grid, d["guess"].data_ptr(), d["dev"].data_ptr(),
                                cfg, d["seeds"].data_ptr(), 0, capi.SCORE_EXACT, po.data_ptr(), co.data_ptr(), st.data_ptr())
            if k > 0:
                ctx.pipeline_flush(1)                       # call k - 1 done before what follows on `stream`
                copies.append(outs[(k - 1) & 1][0].clone())  # ... e.g. the gather of its poses
                outs[(k - 1) & 1][0].zero_()                 # and its buffer may be reused by call k + 1
        ctx.pipeline_flush(0)
        copies.append(outs[(steps - 1) & 1][0].clone())
    ctx.synchronize()
    for c in copies:
        assert torch.equal(c, want)
    ctx.close()


def test_host_buffer_batches_at_depth_two_equal_depth_one():
    """ndtpso_align_pairs (host buffers in and out, synchronous) with batches in flight switched on: its launch goes to a
    lane's stream, so the call has to flush the lanes before it copies the results out and before the next call's uploads
    reuse the staging buffers.  Consecutive calls on different inputs must return what they return at depth 1."""
    from ndtpso_slam_amd import capi, synth
    B, P, I = 160, 30, 20     # more pairs than half the compute units: the pipelined path, not the cluster path
    batches = [synth.make_pairs(B, seed=140 + k) for k in range(4)]
    p0 = batches[0]
    geom = capi.ScanGeom(p0.n_beams, float(p0.angle_min), float(p0.angle_inc), float(p0.range_max), 0.1)
    grid, cfg, dev = capi.Grid(60, 60, 0.5), capi.PSOConfig.make(I, P), (0.1, 0.1, 3.1415e-3)
    ctx = capi.Context(0)

    def run(depth):
        ctx.set_pipeline_depth(depth)
        return [ctx.align_pairs(p.ref_ranges, p.new_ranges, geom, grid, (0, 0, 0), dev, cfg, seeds=p.seeds, mode=capi.SCORE_EXACT)
                for p in batches]

    serial, piped = run(1), run(2)
    for (p1, c1, s1), (p2, c2, s2) in zip(serial, piped):
        assert np.isfinite(p2).all() and (c2 < 0).all()          # not the memset's zeros
        assert np.array_equal(p1, p2) and np.array_equal(c1, c2)
        assert np.array_equal(s1["status"], s2["status"]) and np.array_equal(s1["gbest_updates"], s2["gbest_updates"])
    ctx.set_pipeline_depth(1)
    ctx.close()
