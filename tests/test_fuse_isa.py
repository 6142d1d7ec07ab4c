"""Build-time guard for the one-launch backward (tc_gan_amd/csrc/ssn_fuse.hip).  Its dL/dW update is issued as inline-assembly
MFMAs on accumulator registers, which the compiler's hazard recognizer cannot see: an accumulator move, write or read that the
compiler places right behind such an MFMA would use the tile before the matrix pipe has written it (a first interleaved build
did exactly that in one branch, and 2 % of dL/dW lost an update -- DESIGN 3.7d).  And a single spilled register in the time
loop costs the kernel half its speed (every reload queues behind the HBM prefetch).  Both are properties of the generated
code, so this test compiles the file the way the Makefile does and reads the assembly: no GPU needed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'tc_gan_amd', 'csrc')
UPDATE = re.compile(r'v_mfma_f32_16x16x32_f16 (a\[(\d+):(\d+)\]), v\[\d+:\d+\], v\[\d+:\d+\], \1')


@pytest.fixture(scope='module')
def assembly(tmp_path_factory):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    out = tmp_path_factory.mktemp('fuse') / 'ssn_fuse.s'
    flags = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fno-slp-vectorize']          # csrc/Makefile: CXXFLAGS + FLAGS_ssn_fuse
    subprocess.run([hipcc] + flags + ['-S', '--cuda-device-only', 'ssn_fuse.hip', '-o', str(out)], cwd=CSRC, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def test_no_kernel_of_the_fused_backward_spills(assembly):
    found = re.findall(r'\.name:\s+(\S*gen_backward_fused_kernel\S*).*?\.vgpr_spill_count:\s+(\d+)', assembly, re.S)
    assert len(found) == 6                                # 2N <= 104 / 152 / 208, with and without dL/d ext
    assert all(int(n) == 0 for _, n in found), found


def test_no_compiler_generated_access_to_a_tile_right_behind_its_update(assembly):
    """Within a basic block: no accumulator move / read / write and no memory instruction names a tile register less than
    NEAR instructions behind the inline-assembly MFMA that updates it (the MFMA needs 8 passes; the rescale and the read-out
    wait explicitly and sit in blocks of their own), and no MFMA reads a tile fewer than 3 instructions behind a
    v_accvgpr_write to it (the zeroing of the prologue)."""
    NEAR = 48
    checked = 0
    access = re.compile(r'^(v_accvgpr_(?:mov|write|read)_b32|ds_\w+|global_\w+|buffer_\w+|scratch_\w+)\b(.*)$')
    for kernel in assembly.split('.globl')[1:]:
        if 'gen_backward_fused_kernel' not in kernel.split('\n', 1)[0]:
            continue
        for block in re.split(r'\n\.LBB\d+_\d+:', kernel):
            if not re.search(r';;#ASMSTART\s*\n\s*v_mfma_f32_16x16x32_f16', block):
                continue
            checked += 1
            updated, written = {}, {}
            # (the compiler brackets inline assembly with ;;#ASMSTART / ;;#ASMEND: the chain's MFMAs are builtins, whose
            # hazards it handles itself, and look the same otherwise)
            lines, in_asm = [], False
            for raw in block.split('\n'):
                raw = raw.strip()
                if raw.startswith(';;#ASMSTART'):
                    in_asm = True
                elif raw.startswith(';;#ASMEND'):
                    in_asm = False
                elif raw and not raw.startswith((';', '.')):
                    lines.append((raw, in_asm))
            for k, (line, asm) in enumerate(lines):
                m = UPDATE.match(line) if asm else None
                if m:
                    for r in range(int(m.group(2)), int(m.group(3)) + 1):
                        assert k - written.get(r, -100) >= 3, (line, 'reads a tile right behind a write to it')
                        updated[r] = k
                    continue
                m = access.match(line)
                if not m:
                    continue
                regs = set()
                for lo, hi in re.findall(r'\ba\[(\d+):(\d+)\]', m.group(2)):
                    regs.update(range(int(lo), int(hi) + 1))
                regs.update(int(r) for r in re.findall(r'\ba(\d+)\b', m.group(2)))
                for r in regs:
                    assert k - updated.get(r, -10 ** 6) >= NEAR, (line, 'touches a tile %d instructions behind its MFMA' % (k - updated[r]))
                if line.startswith(('v_accvgpr_write', 'v_accvgpr_mov')):
                    written[int(re.match(r'\S+ a(\d+)', line).group(1))] = k
    assert checked >= 6 * 4 * 2                           # every kernel, every wave, window and main loop
