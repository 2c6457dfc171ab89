"""Round-3 GPU checks: hardened grid barrier of the persistent CG launch (stress: bit-identical over 1 000 runs), the self-healing
path after an aborted persistent launch, and the device-side early-out of solves that run as a chain of launches (wide maps)."""
import os

import pytest
import torch

from test_round2_gpu import _filter_problem

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_persistent_cg_is_bit_identical_over_1000_runs_under_uneven_load():
    """The slab / qbuf exchange of k_cg_run_persistent (write-through payloads drained by EVERY wave before the arrival is counted,
    sc1 reads behind the barrier) must never deliver a stale word: 1 000 runs from the same state give the same bits, with a
    second stream keeping part of the chip busy in bursts so that workgroups reach the barriers unevenly (an idle chip hides stale
    reads, guide 'Test every hand-off under UNEVEN load')."""
    N, c, h, w, Hh, Ww = 80, 96, 30, 54, 480, 854
    mem, opt, wv, g = _filter_problem(N, c, h, w, Hh, Ww, 17, True)
    w0 = wv.detach().clone()
    opt.run((10,))
    ref_w, ref_buf, ref_state = wv.detach().clone(), opt._buf.clone(), opt._state.clone()
    side = torch.cuda.Stream()
    noise = torch.randn(64, 1 << 20, device=DEV)
    bad = 0
    for it in range(1000):
        wv.data.copy_(w0)
        opt.rewind()
        if it % 3 == 0:
            with torch.cuda.stream(side):
                for k in range(1 + it % 5):
                    noise[k * 8:(k + 1) * 8].mul_(1.0000001)       # short streaming kernels on a few dozen CUs
        opt.run((10,))
        if not (torch.equal(wv, ref_w) and torch.equal(opt._buf, ref_buf) and torch.equal(opt._state, ref_state)):
            bad += 1
    torch.cuda.synchronize()
    assert bad == 0, '%d of 1000 persistent runs differ from the first' % bad
    assert not opt.poll_persistent_abort()


def test_aborted_persistent_launch_is_rerun_in_the_multi_kernel_form():
    """a persistent launch that times out must not silently drop the solve (nor every later one).  debug_abort makes the
    first workgroup that waits at a barrier give up: the launch leaves filter / solver state untouched and bumps the sticky abort
    counter; Discriminator.update sees it at the next re-solve frame (or Tracker.run_sequence after its final synchronise), switches to
    the multi-kernel form and RE-RUNS the missed solve.  Later launches start from zeroed barrier words (memset node per launch)."""
    from frtm_vos_amd import ops
    from frtm_vos_amd.model.discriminator import Discriminator
    from frtm_vos_amd.model.optimizer import GaussNewtonCG
    g = torch.Generator().manual_seed(5)
    cin, c, h, w, Hh, Ww = 64, 16, 24, 40, 96, 160
    x0 = torch.relu(torch.randn(3, cin, h, w, generator=g)).to(DEV)
    y0 = torch.zeros(3, 1, Hh, Ww)
    y0[:, 0, 20:60, 30:90] = 1
    frames = [torch.relu(torch.randn(1, cin, h, w, generator=g)).to(DEV) for _ in range(6)]
    m = torch.zeros(1, 1, Hh, Ww)
    m[0, 0, 22:58, 28:88] = 0.9
    m = m.to(DEV)
    cnt = ops.count_above(m.view(1, Hh, Ww))
    outs = {}
    try:
        for mode in ('chain', 'abort_first'):
            GaussNewtonCG.abort_seen_in_process = False
            torch.manual_seed(3)
            d = Discriminator(in_channels=cin, c_channels=c, init_iters=(2, 3), update_iters=(4,), memory_size=8, train_skipping=2,
                              pixel_weighting=dict(method='hinge', tf=0.1), device=DEV, layer='layer4')
            d.init(x0, y0.to(DEV))
            opt = d.update_optimizer
            if mode == 'chain':
                opt.persistent = False
            else:
                assert opt.persistent and opt._persistent_plan() is not None
            filt = []
            for k, ft in enumerate(frames):
                d.apply(ft)
                if mode == 'abort_first' and k == 1:
                    opt.debug_abort = True                         # frame 2 = first re-solve: its persistent launch aborts
                d.update(m, count_dev=cnt)
                opt.debug_abort = False
                if mode == 'abort_first' and k == 1:
                    torch.cuda.synchronize()                        # (so that the mirrored abort counter has landed for the next peek)
                    assert torch.equal(d.filter.weight, filt[-1])   # the aborted launch left the filter alone
                filt.append(d.filter.weight.detach().clone())
            torch.cuda.synchronize()
            d.recover_from_abort()
            outs[mode] = (filt, d.num_persistent_aborts, opt.persistent, d.memory.weights.clone())
        a, b = outs['chain'], outs['abort_first']
        assert a[1] == 0 and b[1] == 1 and b[2] is False and GaussNewtonCG.abort_seen_in_process
        # the solve missed on frame 2 is made up on frame 4 (before that frame's own solve): from then on the filter keeps changing, and
        # the memory (inserts are independent of the solver) is identical
        assert torch.equal(a[3], b[3])
        assert not torch.equal(b[0][3], b[0][2]) and not torch.equal(b[0][5], b[0][3])
        # a NEW target model in this process starts in the multi-kernel form
        torch.manual_seed(3)
        d = Discriminator(in_channels=cin, c_channels=c, init_iters=(2, 3), update_iters=(4,), memory_size=8, train_skipping=2,
                          pixel_weighting=dict(method='hinge', tf=0.1), device=DEV, layer='layer4')
        d.init(x0, y0.to(DEV))
        assert d.update_optimizer.persistent is False
    finally:
        GaussNewtonCG.abort_seen_in_process = False


@pytest.mark.parametrize('shape', [(20, 96, 45, 80, 720, 1280), (12, 96, 68, 120, 1080, 1920)])
def test_device_guard_on_wide_maps_equals_the_host_decision(shape):
    """720p / 1080p memories (80 / 120 columns) do not fit the resident form: their re-solves run as a chain of launches.  With a device-
    resident pixel count the chain is rolled back ON THE DEVICE when the count is below 10 -- same filter, solver state and counters as
    the host-side decision, and no device->host read (round-2 VERDICT missing #4)."""
    N, c, h, w, Hh, Ww = shape
    mem, opt, wv, g = _filter_problem(N, c, h, w, Hh, Ww, 9, True)
    mem2, opt2, wv2, _ = _filter_problem(N, c, h, w, Hh, Ww, 9, True)
    assert opt._persistent_plan() is None and opt.can_guard()
    few, many = torch.tensor([3], dtype=torch.int32, device=DEV), torch.tensor([5000], dtype=torch.int32, device=DEV)
    for guard, solved in ((many, True), (few, False), (many, True)):
        before = (wv.detach().clone(), opt._buf.clone(), opt._state.clone())
        opt.run((10,), guard=guard)
        if solved:
            opt2.run((10,))                                         # the host decided: run
        assert torch.equal(wv, wv2) and torch.equal(opt._buf, opt2._buf) and torch.equal(opt._state, opt2._state)
        if not solved:
            assert torch.equal(wv, before[0]) and torch.equal(opt._buf, before[1]) and torch.equal(opt._state, before[2])
    assert opt.persistent_counts() == (2, 1)
