"""CPU-only checks of the drop-in boundary: the HIP library loads, exports every
symbol include/ssnode_mi355x.h declares, and the product path refuses to run
without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_in_header():
    text = open(os.path.join(ROOT, 'include', 'ssnode_mi355x.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    names = re.findall(r'^\s*(?:int|long|size_t|double|const char \*|const char\*)\s*\*?\s*([a-z_0-9]+)\s*\(', text, flags=re.M)
    return sorted(set(names))


def test_header_declares_reference_exports():
    names = _declared_in_header()
    for ref_symbol in ('solve_dynamics_asym_power_euler', 'solve_dynamics_asym_linear_euler',
                       'solve_dynamics_asym_tanh_euler', 'io_pow', 'io_alin', 'io_atanh',
                       'rate_to_volt', 'dot'):            # nm -D of the reference build
        assert ref_symbol in names


def test_library_loads_and_exports_every_declared_symbol():
    from tc_gan_amd import clib
    names = _declared_in_header()
    assert sorted(clib.DECLARED_SYMBOLS) == names
    for name in names:
        assert hasattr(clib.libssnode, name), name
    assert clib.libssnode.ssn_abi_version() == 1


def test_dynamic_symbol_table_is_exactly_the_header():
    """A drop-in for a ctypes-loaded C library exports its header and nothing else: `nm -D --defined-only` of the product
    library (linked under the version script csrc/gen_export_map.py writes from the header) lists the declared functions --
    the reference's 8 (`nm -D` of a build of tc_gan/ext/ssnode.c) plus the additive batched ABI -- and no C++ launch
    template, kernel stub or HIP registration object."""
    import subprocess
    so = os.path.join(ROOT, 'tc_gan_amd', 'ext', 'libssnode.so')
    out = subprocess.check_output(['nm', '-D', '--defined-only', so]).decode()
    exported = sorted(line.split()[-1].split('@')[0] for line in out.splitlines() if line.strip())
    assert exported == _declared_in_header()


def test_fast_path_table():
    from tc_gan_amd.clib import libssnode
    assert libssnode.ssn_solver_fast_path(200, 1, 4) == 2      # tile kernel
    assert libssnode.ssn_solver_fast_path(204, 8, 4) == 2
    assert libssnode.ssn_solver_fast_path(402, 8, 4) == 0      # falls back to the streaming kernel
    assert libssnode.ssn_solver_fast_path(100, 1, 8) == 2
    assert libssnode.ssn_solver_fast_path(112, 1, 8) == 2      # fp64 beyond 2N = 104: 4 rows per lane, 5-7 waves
    assert libssnode.ssn_solver_fast_path(204, 8, 8) == 2      # the reference's default N = 102 in fp64
    assert libssnode.ssn_solver_fast_path(210, 1, 8) == 0
    assert libssnode.ssn_solver_fast_path(7, 1, 4) == 0         # odd M is invalid


def test_no_cpu_fallback_without_gpu():
    """On a GPU-less host every compute entry point must fail loudly."""
    from tc_gan_amd import clib, ssnode
    if clib.libssnode.ssn_device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(clib.GPUUnavailableError):
        ssnode.fixed_point(np.zeros((2, 2)), [1., 1.], 1., 1.)
    with pytest.raises(clib.GPUUnavailableError):
        ssnode.fixed_points_batch(np.zeros((1, 2, 2)), np.ones((1, 2)), 1., 1.)
    # the raw drop-in symbol reports a library error code (> 900), never a result
    W = np.zeros(4); ext = np.ones(2); r0 = np.zeros(2); r1 = np.zeros(2)
    rc = clib.libssnode.solve_dynamics_asym_tanh_euler(
        1, W.ctypes.data_as(clib.double_ptr), ext.ctypes.data_as(clib.double_ptr), 1., 1.,
        r0.ctypes.data_as(clib.double_ptr), r1.ctypes.data_as(clib.double_ptr),
        .01, .002, 8e-4, 10, 1e-5, 200., 1000.)
    assert rc > 900
    assert np.all(r0 == 0)


def test_product_package_never_imports_oracle():
    """The oracle is test infrastructure: nothing under tc_gan_amd/ may reference it."""
    pkg = os.path.join(ROOT, 'tc_gan_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', text, flags=re.M), f
                assert 'liboracle' not in text and '_ref/' not in text, f
