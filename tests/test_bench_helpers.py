"""CPU: the pure helpers of bench.py that the judged line is computed with - SURVEY 8(d)'s algorithmic bytes and the true bound of
a resident-pass round (last review, item 1c) - and the staging directory of the reference archive (advisor, round 5)."""
import os
import stat

import bench


def test_algorithmic_bytes_are_surveys_figures():
    assert bench.bytes_fwd(32) == 21396464 and bench.bytes_fwd(128) == 29528816 and bench.bytes_fwd(1024) == 105430768
    assert bench.bytes_fwd(32, skin_topk=4) == 20955504 and bench.bytes_fwd(128, skin_topk=4) == 29087856
    assert bench.bytes_fwd(32, skin_topk=4, half_basis=True) == 20955504 - 8557380 - 413400


def test_true_bound_of_a_resident_round():
    """max(bytes moved / 8 TB/s, MFMA issue of the busiest SIMD at 2.4 GHz, fp32 vector flops / peak of the CUs held)."""
    b32 = bench.true_bound(32, 1, 216)
    assert b32['which'] == 'mfma' and abs(b32['us'] - 2 * 21 * 32 / 2400.0) < 1e-9             # two 21-MFMA chains on the busiest SIMD
    assert abs(b32['components_us']['hbm'] - 84712 * 32 / 8e12 * 1e6) < 1e-3
    b128 = bench.true_bound(128, 3, 108)
    assert b128['which'] == 'mfma' and abs(b128['us'] - 4 * 3 * 21 * 32 / 2400.0) < 1e-9       # four chunks x three chains per SIMD
    assert abs(bench.true_bound(128, 3, 108, half_basis=True)['us'] - 4 * 3 * 14 * 32 / 2400.0) < 1e-9
    assert abs(b128['components_us']['valu'] - 6890 * 128 * 130 / (157.3e12 * 108 / 256) * 1e6) < 1e-3
    # many problems on few CUs: the vector work becomes the floor
    assert bench.true_bound(128, 3, 16)['which'] == 'valu'


def test_the_staged_reference_is_unpacked_where_only_this_user_can_write():
    from oracle import ref_import as ri
    if not ri.STAGED:
        # build container: /root/reference is mounted and used as is; the staging path is exercised by resolving it explicitly
        if not os.path.isfile(ri._STAGE):
            return
        orig = os.path.isfile
        os.path.isfile = lambda p: False if p == '/root/reference/code/utils/fitting.py' else orig(p)
        try:
            root = ri._resolve_root()
        finally:
            os.path.isfile = orig
    else:
        root = ri.REF_ROOT
    assert os.path.isfile(os.path.join(root, 'code', 'utils', 'fitting.py')) and os.path.isfile(os.path.join(root, '.complete'))
    st = os.stat(os.path.dirname(root))
    assert st.st_uid == os.getuid() and not (stat.S_IMODE(st.st_mode) & 0o077), oct(st.st_mode)
