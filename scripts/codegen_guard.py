#!/usr/bin/env python3
"""Guards on the GENERATED code of the persistent ping-pong kernels: disassembles the gfx950 code objects of the in-tree build (CPU only, llvm-objdump).

The kernels rely on properties hipcc does not promise: (1) no scratch - a scratch reload waits on vmcnt, i.e. drains every LDS-DMA in flight (EXPERIMENTS.md: seen every time
one more register was live across the tile loop of a 256-VGPR kernel); (2) the tile queue's ticket (conv_pp_common.hpp tq_draw) comes back ASYNCHRONOUSLY into a VGPR the compiler
believes is defined at once: between the global_atomic_add and the ds_write that posts the ticket nothing may read, copy, spill or reuse that register; (3) the five wait states
between a VALU write of the counter address (v_readlane from a spill lane) and the atomic - inside an asm block the hazard recogniser does not help (round 5: a GPU fault).

Run by the Makefile after every link (`make` fails when a guard fails or llvm-objdump is missing - ADVICE r5: the only thing between the tile queue and that fault must not be a
test that skips), by __graft_entry__.build() through it, and by tests/test_codegen_guards.py.  Exit code 0 = all guards hold."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mdeical_image_segmentation_amd", "csrc")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
QUEUE_KERNELS = {"conv_pp.o": "conv_ppc_kernel", "conv_ppd.o": "conv_ppd_kernel", "conv3d_pp.o": "conv3d_ppc_kernel"}


def disasm(obj):
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copy(os.path.join(CSRC, obj), tmp)
        subprocess.run([OBJDUMP, "--offloading", obj], cwd=tmp, check=True, capture_output=True)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        assert len(co) == 1, os.listdir(tmp)
        out = subprocess.run([OBJDUMP, "-d", co[0]], cwd=tmp, check=True, capture_output=True, text=True).stdout
    kernels, name = {}, None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1)
            kernels[name] = []
        elif name is not None and line.strip():
            kernels[name].append(line.split("//")[0].strip())
    return kernels


def _touches(instr, n):
    if re.search(r"\bv%d\b" % n, instr):
        return True
    return any(int(a) <= n <= int(b) for a, b in re.findall(r"v\[(\d+):(\d+)\]", instr))


def check_object(obj):
    """raises AssertionError naming the kernel and the instruction; returns the number of ticket draws checked"""
    kernels = {k: v for k, v in disasm(obj).items() if QUEUE_KERNELS[obj] in k}
    assert kernels, f"no {QUEUE_KERNELS[obj]} in {obj}"
    drawn = 0
    for name, ins in kernels.items():
        assert not any(i.startswith("scratch_") for i in ins), f"{name}: scratch instructions (a spill in a kernel whose waits are counted by hand)"
        for i, t in enumerate(ins):
            m = re.match(r"global_atomic_add v(\d+), v\d+, v\d+, s\[", t)
            if not m:
                continue
            drawn += 1
            assert ins[i - 1].startswith("s_nop 4"), f"{name}: the draw is not preceded by its wait states: {ins[i - 1]}"
            n = int(m.group(1))
            nxt = next((u for u in ins[i + 1:] if _touches(u, n)), None)
            assert nxt is not None and nxt.startswith("ds_write_b32") and nxt.endswith(f"v{n}"), f"{name}: v{n} (a ticket in flight) is touched by `{nxt}` before it is posted"
    assert drawn >= 2, f"{obj}: the tile queue is not compiled in"
    return drawn


# The all-DMA fp32 kernels (conv3d_f32.hip, wgrad_f32.hip): a ds_read that starts within a few dozen cycles of the barrier that published ANOTHER wave's LDS-DMA can still see the
# old bytes (wgrad_pp.hip).  Those kernels rely on distance, not on a documented wait: behind every s_barrier either an s_sleep or at least four f32 MFMAs (4 x 32
# cycles) come before the first LDS read.  ADVICE r5: a different schedule (another NF, a compiler upgrade) could close that gap silently - this check pins it.
MARGIN_KERNELS = {"conv3d_f32.o": "conv3d_f32_kernel", "wgrad_f32.o": "wgrad_f32_stream_kernel"}


def check_margins(obj):
    kernels = {k: v for k, v in disasm(obj).items() if MARGIN_KERNELS[obj] in k}
    assert kernels, f"no {MARGIN_KERNELS[obj]} in {obj}"
    n = 0
    for name, ins in kernels.items():
        for i, t in enumerate(ins):
            if not t.startswith("s_barrier"):
                continue
            mfma = sleep = 0
            for u in ins[i + 1:]:
                if u.startswith("ds_read"):
                    break
                mfma += u.startswith("v_mfma")
                sleep += u.startswith("s_sleep")
            assert sleep >= 1 or mfma >= 4, f"{name}: only {mfma} MFMAs and no s_sleep between a barrier (instruction {i}) and the next LDS read"
            n += 1
    return n


def main():
    if not os.path.exists(OBJDUMP):
        print(f"codegen guard: {OBJDUMP} is missing - the guard cannot run, and the build must not pass without it", file=sys.stderr)
        return 2
    rc = 0
    for obj in sorted(QUEUE_KERNELS):
        if not os.path.exists(os.path.join(CSRC, obj)):
            print(f"codegen guard: {obj} has not been built", file=sys.stderr)
            rc = 2
            continue
        try:
            n = check_object(obj)
            print(f"codegen guard: {obj}: {n} ticket draws, no scratch - ok")
        except AssertionError as e:
            print(f"codegen guard FAILED: {obj}: {e}", file=sys.stderr)
            rc = 1
    for obj in sorted(MARGIN_KERNELS):
        if not os.path.exists(os.path.join(CSRC, obj)):
            print(f"codegen guard: {obj} has not been built", file=sys.stderr)
            rc = 2
            continue
        try:
            print(f"codegen guard: {obj}: {check_margins(obj)} barriers, each followed by an s_sleep or >= 4 MFMAs before the next LDS read - ok")
        except AssertionError as e:
            print(f"codegen guard FAILED: {obj}: {e}", file=sys.stderr)
            rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
