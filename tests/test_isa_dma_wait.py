"""The hand-counted `s_waitcnt vmcnt(N)` of conv_f32_dma_kernel, checked on the EMITTED code.

The kernel (sparsefusion_amd/csrc/spconv_f32_ordered.hip) writes gathered rows into LDS with
`buffer_load_dwordx4 ... lds` two phases ahead and ends every phase with a counted wait in front of
a raw `s_barrier`.  The wait is correct only if, on every path, at least N vector-memory
operations are issued between a row DMA and the wait of the FOLLOWING phase (vmcnt retires in
order, so `vmcnt(N)` leaves only the youngest N operations outstanding).  Nothing in the language
pins that: a compiler that merges, hoists or sinks one of those loads turns the wait into a race
which parity tests catch only by luck.  So this test disassembles the built object and walks the
control-flow graph of every instantiation: from each row DMA, along every path, a wait
`vmcnt(X)` with X <= (vector-memory loads issued since the DMA) proves the DMA has landed; a path
that reaches the second `s_barrier` after the DMA without such a wait is a failure.

CPU only: runs on the in-tree object file hipcc cross-compiled (no GPU needed).
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, 'sparsefusion_amd', 'csrc', 'spconv_f32_ordered.o')
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'

LINE = re.compile(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):')
VMCNT = re.compile(r'vmcnt\((\d+)\)')


def disassemble():
    from sparsefusion_amd import _lib
    if not os.path.exists(OBJ):
        _lib.build()
    tmp = tempfile.mkdtemp(prefix='sf_isa_')
    try:
        o = os.path.join(tmp, 'o.o')
        shutil.copy(OBJ, o)
        subprocess.check_call([OBJDUMP, '--offloading', o], cwd=tmp, stdout=subprocess.DEVNULL)
        co = [f for f in os.listdir(tmp) if 'gfx950' in f]
        assert len(co) == 1, os.listdir(tmp)
        return subprocess.check_output([OBJDUMP, '-d', os.path.join(tmp, co[0])], text=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def kernels(text):
    """{mangled name: [(addr, mnemonic, operands)]} of the dma kernels."""
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
        if m:
            cur = out.setdefault(m.group(1), []) if 'conv_f32_dma_kernel' in m.group(1) else None
            continue
        if cur is None:
            continue
        m = LINE.match(line)
        if m:
            ops = m.group(2)
            cur.append((int(m.group(3), 16), m.group(1), ops))
    return out


def successors(ins, index_of):
    """CFG edges of the instruction list (indices)."""
    succ = []
    for i, (addr, mn, ops) in enumerate(ins):
        s = []
        if mn.startswith('s_cbranch') or mn == 's_branch':
            off = int(ops.split()[0])
            if off >= 0x8000:
                off -= 0x10000
            tgt = addr + 4 + 4 * off
            assert tgt in index_of, (mn, ops, hex(addr))
            s.append(index_of[tgt])
        if mn not in ('s_branch', 's_endpgm') and i + 1 < len(ins):
            s.append(i + 1)
        succ.append(s)
    return succ


def is_vmem_load(mn):
    return mn.startswith(('global_load', 'buffer_load', 'flat_load', 'scratch_load'))


def check_kernel(name, ins):
    index_of = {a: i for i, (a, _, _) in enumerate(ins)}
    succ = successors(ins, index_of)
    dmas = [i for i, (_, mn, ops) in enumerate(ins) if mn.startswith('buffer_load') and ops.rstrip().endswith('lds')]
    assert dmas, f'{name}: no LDS-DMA instruction found'
    cap = 64
    worst = None
    for d in dmas:
        # state: (instruction about to execute, loads issued since the DMA, barriers passed)
        start = [(s, 0, 0) for s in succ[d]]
        seen = set(start)
        stack = list(start)
        while stack:
            i, m, nb = stack.pop()
            _, mn, ops = ins[i]
            if mn == 's_waitcnt':
                w = VMCNT.search(ops)
                if w and int(w.group(1)) <= m:
                    continue                      # the DMA has landed on this path
            if mn == 's_barrier':
                nb += 1
                if nb == 2:
                    raise AssertionError(
                        f'{name}: the DMA at {ins[d][0]:#x} can reach the second barrier after it '
                        f'({ins[i][0]:#x}) with only {m} vector-memory loads issued behind it and no '
                        f'wait that covers it')
            if mn == 's_endpgm':
                continue
            if is_vmem_load(mn):
                m = min(m + 1, cap)
            for s in succ[i]:
                st = (s, m, nb)
                if st not in seen:
                    seen.add(st)
                    stack.append(st)
        worst = d
    return len(dmas)


@pytest.fixture(scope='module')
def dma_kernels():
    if not os.path.exists(OBJDUMP):
        pytest.skip('llvm-objdump not in this image')
    ks = kernels(disassemble())
    assert ks, 'no conv_f32_dma_kernel instantiation in the object file'
    return ks


def test_counted_wait_covers_every_dma(dma_kernels):
    for name, ins in dma_kernels.items():
        n = check_kernel(name, ins)
        assert n >= 3, (name, n)                   # three unrolled phases, >= 1 DMA each


def test_counted_wait_is_the_documented_constant(dma_kernels):
    """Every raw barrier of the phase loop is preceded by `s_waitcnt vmcnt(2*IPW + 2*NBL)` (the
    template arguments give IPW and NBL); the other barriers of the kernel follow a full drain."""
    for name, ins in dma_kernels.items():
        cin, cout, cg, waves, _ = [int(x) for x in re.findall(r'Li(\d+)E', name)[:5]]
        ipw = 64 // (1024 // (cin * 4)) // waves
        nbl = (cout // 16 // cg) * (cin // 16)
        want = 2 * ipw + 2 * nbl
        counted = 0
        for i, (_, mn, ops) in enumerate(ins):
            if mn != 's_barrier':
                continue
            j = i - 1
            while j >= 0 and ins[j][1] != 's_waitcnt':
                j -= 1
            w = VMCNT.search(ins[j][2]) if j >= 0 else None
            assert w is not None and i - j <= 6, (name, hex(ins[i][0]))
            if int(w.group(1)) == want:
                counted += 1
            else:
                assert int(w.group(1)) == 0, (name, hex(ins[i][0]), ins[j][2])
        assert counted == 3, (name, counted, want)


def test_the_check_catches_a_short_wait(dma_kernels):
    """Self-test of the walker: the same code with the counted waits weakened by one must fail."""
    name, ins = next(iter(dma_kernels.items()))
    cin, cout, cg, waves, _ = [int(x) for x in re.findall(r'Li(\d+)E', name)[:5]]
    want = 2 * (64 // (1024 // (cin * 4)) // waves) + 2 * (cout // 16 // cg) * (cin // 16)
    bad = [(a, mn, ops.replace(f'vmcnt({want})', f'vmcnt({want + 1})') if mn == 's_waitcnt' else ops)
           for a, mn, ops in ins]
    with pytest.raises(AssertionError):
        check_kernel(name, bad)
