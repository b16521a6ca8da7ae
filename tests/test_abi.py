"""The C-ABI shared library loads without a GPU and exports every symbol include/dgnn_spmm.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'dgnn_spmm.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(dgnn_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, 'dist-gnn_amd', 'lib', 'libdgnn_spmm.so'))
    names = _declared()
    assert {'dgnn_spmm_csr_f32', 'dgnn_spmm_coo_f32', 'dgnn_coo2csr_i32', 'dgnn_spmm_plan_f32',
            'dgnn_pad_rows_f32', 'dgnn_last_error', 'dgnn_version'} <= set(names)
    for n in names:
        assert hasattr(lib, n), f'{n} declared in the header but not exported'
    lib.dgnn_version.restype = ctypes.c_int
    assert lib.dgnn_version() >= 100


def test_argument_checks_need_no_gpu():
    lib = ctypes.CDLL(os.path.join(ROOT, 'dist-gnn_amd', 'lib', 'libdgnn_spmm.so'))
    lib.dgnn_last_error.restype = ctypes.c_char_p
    rc = lib.dgnn_coo2csr_i32(None, ctypes.c_int64(5), ctypes.c_int32(3), None, None)
    assert rc == 1 and b'null' in lib.dgnn_last_error()
    rc = lib.dgnn_spmm_plan_f32(None, None, None, 1, 1, None, ctypes.c_int64(4), None, ctypes.c_int64(4), None,
                                ctypes.c_int64(4), 4, ctypes.c_float(1), ctypes.c_float(1), None)
    assert rc == 1


def test_binding_rejects_cpu_tensors():
    import pytest
    import torch
    import spmm_cpp
    i = torch.zeros(1, dtype=torch.int32)
    with pytest.raises(RuntimeError, match='no CPU path'):
        spmm_cpp.spmm_cusparse(i, i, torch.ones(1), 1, 1, torch.ones(1, 1), torch.ones(1, 1), 1, 1)
    assert spmm_cpp.padded_width(602) == 608 and spmm_cpp.padded_width(41) == 64 and spmm_cpp.padded_width(256) == 256
    assert spmm_cpp.default_stage_count(232965, 232965, 114848857, 602) == 4
    assert spmm_cpp.default_stage_count(232965, 232965, 114848857, 256) == 4
    assert spmm_cpp.default_stage_count(232965, 232965, 114848857, 41) == 2
    assert spmm_cpp.default_stage_count(29121, 29121, 1819992, 602) == 1          # a block of the 8-rank partition
    assert spmm_cpp.default_stage_count(2449029, 2449029, 126167309, 100) == 2    # products: capped by run length
    assert spmm_cpp.default_piece_len(114848857, 4) == 4096 and spmm_cpp.default_piece_len(1819992, 1) == 256
    assert spmm_cpp.default_piece_len(1000, 1) == 128


def test_binding_signatures_match_the_header():
    """The ctypes stub declares every entry point of the header with the same number of arguments."""
    import spmm_cpp
    txt = open(os.path.join(ROOT, 'include', 'dgnn_spmm.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    txt = re.sub(r'//[^\n]*', '', txt)
    protos = {m.group(1): m.group(2) for m in re.finditer(r'\b(dgnn_[a-z0-9_]+)\s*\(([^)]*)\)\s*;', txt)}
    assert set(protos) - {'dgnn_last_error'} == set(spmm_cpp._SIGNATURES)
    for name, args in spmm_cpp._SIGNATURES.items():
        decl = protos[name].strip()
        count = 0 if decl in ('', 'void') else decl.count(',') + 1
        assert count == len(args), f'{name}: header has {count} arguments, the stub {len(args)}'


def test_missing_library_is_a_loud_error_not_an_import_error(tmp_path):
    """Without libdgnn_spmm.so the stub must raise (RuntimeError, not ImportError: the reference's
    `try: from spmm_cpp import ...; except ImportError:` at gcn.py:9-13 would otherwise fall back to addmm_ silently)."""
    import shutil
    import subprocess
    import sys
    pkg = tmp_path / 'pkg'
    shutil.copytree(os.path.join(ROOT, 'dist-gnn_amd', 'spmm_cpp'), pkg / 'spmm_cpp')      # the stub without lib/
    code = ('import sys; sys.path.insert(0, %r)\n'
            'try:\n    import spmm_cpp\nexcept ImportError as e:\n    print("IMPORTERROR", e)\n'
            'except RuntimeError as e:\n    print("RUNTIMEERROR", e)\nelse:\n    print("IMPORTED")\n' % str(pkg))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120).stdout
    assert out.startswith('RUNTIMEERROR') and 'no CPU fallback' in out, out
