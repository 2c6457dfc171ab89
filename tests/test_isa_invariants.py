"""The fused Winograd kernel (csrc/conv_wino.hip) counts its load queue by hand (s_waitcnt vmcnt(N) with N = loads that may stay in flight).  The counts
only mean what the source says if the compiler keeps the loads unconditional, in source order, and adds no loads of its own (register spills are
scratch loads: they count in vmcnt too).  Round 5 found all three violated at one time or another -- each time silently, as a slower or a racy kernel.
This test compiles the file to ISA (no GPU needed) and checks the invariants the header comment of conv_wino.hip lists."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = '/opt/rocm/bin/hipcc'


@pytest.fixture(scope='module')
def wino_isa():
    if not os.path.exists(HIPCC):
        pytest.skip('no hipcc')
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'wino.s')
        subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', out,
                        os.path.join(ROOT, 'frtm-vos_amd', 'csrc', 'conv_wino.hip')], check=True, capture_output=True, cwd=d)
        return open(out).read()


def _loop_body(L):
    """Lines of the K loop of a function listing (the innermost loop that holds MFMAs): from its header to its back edge."""
    found = []
    for h in [i for i, l in enumerate(L) if 'Inner Loop Header' in l]:
        tag = re.search(r'(BB\d+_\d+)', L[h]).group(1)
        inside = [i for i, l in enumerate(L) if ('in Loop: Header=%s ' % tag) in l + ' ']
        end = (inside[-1] if inside else h) + 1
        while end < len(L) and not re.match(r'\.LBB\d+_\d+:', L[end]):
            end += 1
        body = L[h:end]
        cut = max(i for i, l in enumerate(body) if re.search(r'\ss_cbranch', l))  # the back edge: what follows it in the last block is past the loop
        body = body[:cut + 1]
        if any('v_mfma' in l for l in body):
            found.append(body)
    assert len(found) == 1, 'one K loop expected'
    return found[0]


def _function(isa, mangled):
    m = re.search(r'^%s:.*?^\.Lfunc_end' % re.escape(mangled), isa, flags=re.S | re.M)
    assert m, mangled
    return m.group(0).split('\n')


@pytest.mark.parametrize('fn,tall,waves,nr', [(2, 0, 3, 6), (2, 1, 3, 6), (1, 0, 5, 4)])
def test_winograd_kernel_load_queue_is_what_the_source_counts(wino_isa, fn, tall, waves, nr):
    name = '_Z14k_conv3x3_winoILi%dELi%dELi%dEEv10ConvParams' % (fn, tall, waves)
    L = _function(wino_isa, name)
    body = _loop_body(L)
    # 1. nothing spilled inside the loop (the 32-tile forms must not spill at all)
    assert not [l for l in body if 'scratch_' in l], 'scratch access inside the K loop'
    if fn == 2:
        m = re.search(r'\.name:\s+%s\s.*?\.vgpr_spill_count:\s+(\d+)' % re.escape(name), wino_isa, flags=re.S)
        assert m and int(m.group(1)) == 0
    # 2. the loop body = two chunks, each: 4 weight loads + NR patch loads (LDS-DMA), all unconditional (no branch inside the body), weights first
    branches = [i for i, l in enumerate(body) if re.search(r'\ss_cbranch', l)]
    last_mfma = max(i for i, l in enumerate(body) if 'v_mfma' in l)
    assert len(branches) == 1 and branches[0] > last_mfma - 40, 'a branch inside the K loop body (besides its back edge): loads under a condition break the counts'
    loads = [('A' if 'dwordx4' in l else 'R') for l in body if re.search(r'\sbuffer_load_dword', l)]
    chunk = ['A'] * 4 + ['R'] * nr
    assert loads == chunk * 2, loads
    # 3. the barrier of a chunk waits for the previous patch only: vmcnt(NR + 4) directly in front of each s_barrier; the MFMAs wait for their weights
    #    with exact counts (first use: 4 + NR + 3 loads may be newer), never for everything
    bars = [i for i, l in enumerate(body) if 's_barrier' in l]
    assert len(bars) == 2
    for b in bars:
        prev = [l for l in body[max(0, b - 12):b] if 's_waitcnt vmcnt' in l]
        assert prev and re.search(r'vmcnt\((\d+)\)', prev[-1]).group(1) == str(nr + 4), (prev, nr + 4)
    waits = [int(re.search(r'vmcnt\((\d+)\)', l).group(1)) for l in body if 's_waitcnt vmcnt' in l]
    assert min(waits) == nr + 4, waits                     # nothing inside the loop waits for more than the chunk barrier does
    assert waits.count(nr + 4) == 2
    assert sum(1 for l in body if 'v_mfma_f32_16x16x4_f32' in l) == 2 * 16 * fn


@pytest.fixture(scope='module')
def igemm_isa():
    if not os.path.exists(HIPCC):
        pytest.skip('no hipcc')
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'igemm.s')
        subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-o', out,
                        os.path.join(ROOT, 'frtm-vos_amd', 'csrc', 'conv_igemm.hip')], check=True, capture_output=True, cwd=d)
        return open(out).read()


@pytest.mark.parametrize('inst,mfmas', [('Li64ELi64ELi2ELi4ELi1ELi32E', 16), ('Li32ELi64ELi1ELi4ELi1ELi32E', 16)])
def test_gemm_kernel_k_loop_carries_no_address_arithmetic(igemm_isa, inst, mfmas):
    """Every VALU instruction takes ~4 cycles of matrix-pipe time (tools/mfma_valu_probe.hip): the K loop of k_conv_igemm's 1x1 forms (csrc/conv_igemm.hip)
    is unrolled by two so that LDS buffer, k-step and fragment offsets are immediates and the operand loads take their row offsets through the scalar
    offset.  Guard: per chunk at most 3 VALU instructions besides the MFMAs on the path every chunk takes (the loop body as compiled contains the
    checked tail form of the loads as a second branch: its instructions are not counted)."""
    name = '_Z12k_conv_igemmI%sEv10ConvParams' % inst
    L = _function(igemm_isa, name)
    body = _loop_body(L)
    assert sum(1 for l in body if 'v_mfma_f32_16x16x4_f32' in l) == 2 * mfmas
    assert not [l for l in body if 'scratch_' in l]
    # fragment reads: ds_read_b32 with immediate offsets on two base registers, no ds_read2 pairs, no v_add in front of them
    reads = [l for l in body if re.search(r'\sds_read', l)]
    assert reads and all('ds_read_b32' in l and 'offset:' in l for l in reads), reads[:3]
    assert len({re.search(r'ds_read_b32 v\d+, (v\d+)', l).group(1) for l in reads}) == 2
    # the straight path: blocks of the body that are not the tail-form branch (recognised by its v_cndmask / v_cmp offset selects)
    valu = [l for l in body if re.match(r'\s+v_', l) and 'v_mfma' not in l]
    tail_form = [l for l in valu if re.search(r'v_cndmask|v_cmp|v_bfrev|v_subrev', l)]
    straight = [l for l in valu if l not in tail_form]
    assert len(straight) <= 2 * 6, straight
