"""The emitted gfx950 ISA of the one-launch factorisation keeps its 16-byte stores' data registers untouched for long enough.

A 16-byte buffer store reads its data registers over several cycles after it has issued; a VALU / load instruction that
rewrites one of them too early corrupts the store (documented for gfx9 as "VMEM store of more than 64 bits followed by a write
of the data VGPRs: wait states"; LLVM inserts them only when the store's soffset is not a register).  In round 4 the chain's
panel-tile store (csrc/chol.hip store_pt_sc1) lost data that way when other processes' waves shared the CU -- silent wrong
factors, a few launches per thousand.  The source now keeps every value in registers of its own and waits; nothing obliges
a future compiler to keep that shape, so this test reads the shipped library's code: in potrf_mega_kernel no instruction
may write a data register of a buffer_store_dwordx4 within the next WAIT wait states."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'dgp_amd', 'libdgp_amd.so')
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
WAIT = 2        # wait states the ISA asks for between the store and a write of its data registers
WAIT_PT = 8     # ... and what the chain's panel-tile stores (eight in a row, sc1) are given at least: the round-4 failure needed more than 2


def device_objects(path):
    """the gfx950 code objects inside the library's .hip_fatbin section (one clang offload bundle per translation unit)"""
    blob = open(path, 'rb').read()
    out, at = [], 0
    while True:
        at = blob.find(MAGIC, at)
        if at < 0:
            break
        n, = struct.unpack_from('<Q', blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if 'gfx950' in triple and size:
                out.append(blob[at + off:at + off + size])
        at += len(MAGIC)
    return out


def kernel_isa(tmp_path, symbol):
    for i, obj in enumerate(device_objects(LIB)):
        if symbol.encode() not in obj:
            continue
        f = tmp_path / ('dev%d.o' % i)
        f.write_bytes(obj)
        txt = subprocess.run([OBJDUMP, '-d', '--mcpu=gfx950', '--no-show-raw-insn', str(f)], capture_output=True, text=True, check=True).stdout
        m = re.search(r'^[0-9a-f]+ <%s>:\n(.*?)(?=^\s*$|^[0-9a-f]+ <)' % re.escape(symbol), txt, re.S | re.M)
        if m:
            return [ln.split('//')[0].strip() for ln in m.group(1).splitlines() if ln.strip() and not ln.strip().startswith(';')]
    return None


def regs(op):
    """VGPR numbers named by one operand: v12 or v[12:15]"""
    m = re.fullmatch(r'v(\d+)', op)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def written(ins):
    """VGPRs an instruction writes (first operand of VALU / MFMA / load instructions; both of a swap)"""
    parts = ins.replace(',', ' ').split()
    op, args = parts[0], parts[1:]
    if not args or op.startswith(('s_', 'buffer_store', 'global_store', 'flat_store', 'ds_write', 'ds_store', 'scratch_store', 'buffer_wbl2', 'buffer_inv', 'buffer_atomic', 'global_atomic')):
        return set()
    if 'swap' in op:
        return regs(args[0]) | (regs(args[1]) if len(args) > 1 else set())
    return regs(args[0])


def wait_states(ins):
    m = re.match(r's_nop (\d+)', ins)
    return int(m.group(1)) + 1 if m else 1


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump of the ROCm toolchain not found')
def test_16_byte_stores_keep_their_data_registers(tmp_path):
    isa = kernel_isa(tmp_path, '_Z17potrf_mega_kernel8MegaArgs')
    assert isa and len(isa) > 5000, 'potrf_mega_kernel not found in the library'
    stores = [i for i, ins in enumerate(isa) if ins.startswith('buffer_store_dwordx4')]
    assert len(stores) >= 16, len(stores)
    runs = 0
    for idx, i in enumerate(stores):
        ops = isa[i].replace(',', ' ').split()
        data = regs(ops[1])
        assert len(data) == 4, isa[i]
        # soffset (the operand behind the resource): a literal, never an SGPR -- LLVM inserts its own wait states only then
        soff = ops[ops.index(next(o for o in ops if o.startswith('s['))) + 1]
        assert not re.fullmatch(r's\d+|m0', soff), 'soffset of %s is a register: LLVM\'s hazard recogniser skips this store' % isa[i]
        # the panel-tile stores: eight consecutive sc1 stores (csrc/chol.hip store_pt_sc1)
        in_run = all(j < len(isa) and isa[j].startswith('buffer_store_dwordx4') and 'sc1' in isa[j] for j in range(i, i + 1)) and \
            sum(1 for j in range(max(0, i - 7), min(len(isa), i + 8)) if isa[j].startswith('buffer_store_dwordx4')) >= 8
        need = WAIT_PT if in_run else WAIT
        ws, j = 0, i + 1
        while ws < need and j < len(isa):
            hit = written(isa[j]) & data
            assert not hit, ('%s (instruction %d) rewrites v%s %d wait states behind %s' % (isa[j], j, sorted(hit), ws, isa[i]))
            ws += wait_states(isa[j])
            if isa[j].startswith(('s_endpgm', 's_branch', 's_cbranch', 's_setpc')):
                break
            j += 1
        runs += in_run
    assert runs >= 8, 'the chain\'s eight-store run was not recognised: the test no longer watches what it was written for'


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason='llvm-objdump of the ROCm toolchain not found')
def test_lds_dma_is_only_ever_waited_for_with_vmcnt_zero(tmp_path):
    """The Matern pair kernel stages its records with LDS-DMA loads (global_load_lds_dwordx4).  Those do not retire in order with register
    loads: a build that had register loads in flight together with them and waited with s_waitcnt vmcnt(N > 0) -- the compiler's way of waiting for
    the OLDER of several loads -- gave wrong sums for most test points (round 5, profiles/r05_pair_kernel_staging.txt, block 11).  The shipped
    kernel is right because every wait behind its first DMA load is a vmcnt(0); this test keeps it that way, in every variant the library holds, and
    checks that the kernel drains its DMA ahead of s_endpgm."""
    seen = 0
    for sym in ('_Z18linkgp_Jsep_kernelILi2ELb0EEv8LinkArgs', '_Z18linkgp_Jsep_kernelILi0ELb0EEv8LinkArgs', '_Z18linkgp_Jsep_kernelILi2ELb1EEv8LinkArgs'):
        isa = kernel_isa(tmp_path, sym)
        assert isa and len(isa) > 1000, sym + ' not found in the library'
        first = next(i for i, ins in enumerate(isa) if ins.startswith('global_load_lds'))
        for i in range(first + 1, len(isa)):
            m = re.search(r'vmcnt\((\d+)\)', isa[i]) if isa[i].startswith('s_waitcnt') else None
            if m:
                assert int(m.group(1)) == 0, '%s: %s (instruction %d) waits for part of the vector-memory operations with LDS-DMA loads possibly in flight' % (sym, isa[i], i)
        # every barrier behind the first DMA load has the issuing wave's vector-memory operations drained ahead of it: the other waves read the
        # DMA's LDS destination right behind the barrier, and a workgroup-scope fence alone orders lgkmcnt traffic only (the in-loop
        # barrier was right by the compiler's placement of its own wait, not by the source).  "Ahead" = a vmcnt(0) wait with no vector-memory
        # instruction and no branch target between it and the s_barrier.
        bars = [i for i in range(first + 1, len(isa)) if isa[i].startswith('s_barrier')]
        assert len(bars) >= 2, sym + ': the in-loop and the exit barrier were expected behind the first DMA load'
        for b in bars:
            j = b - 1
            while j > first and not (isa[j].startswith('s_waitcnt') and 'vmcnt(0)' in isa[j]):
                assert not isa[j].startswith(('global_', 'buffer_', 'flat_', 'scratch_')), \
                    '%s: %s (instruction %d) sits between the last vmcnt(0) and the barrier at %d' % (sym, isa[j], j, b)
                assert not isa[j].startswith(('s_branch', 's_cbranch', 's_setpc', 's_endpgm')), \
                    '%s: control flow (%s, instruction %d) between the last vmcnt(0) and the barrier at %d' % (sym, isa[j], j, b)
                j -= 1
            assert j > first, '%s: no vmcnt(0) ahead of the barrier at instruction %d' % (sym, b)
            assert b - j <= 4, '%s: the vmcnt(0) nearest to the barrier at %d is %d instructions ahead of it' % (sym, b, b - j)
        end = max(i for i, ins in enumerate(isa) if ins.startswith('s_endpgm'))
        last_dma = max(i for i, ins in enumerate(isa) if ins.startswith('global_load_lds'))
        assert any(isa[i].startswith('s_waitcnt') and 'vmcnt(0)' in isa[i] for i in range(last_dma + 1, end)), sym + ': no vmcnt(0) between the last DMA load and s_endpgm'
        seen += 1
    assert seen == 3
