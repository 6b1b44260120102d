"""Cross-implementation parity against GENUINE HElib 2.2.0, for whoever has it (VERDICT r4 missing #3).

tools/helib_dump/dump.cpp, built against real HElib, writes the chain, the roots NTL's PRG produced, keys, two
ciphertexts, their product, the level-2 product and a rotation; with HELIB_DUMP_DIR pointing at such a directory these
tests register the dumped roots, replay the operations with this repository's host logic and compare the chain, the
prime-set decisions, intFactor and EVERY WORD of every ciphertext part -- on the oracle (CPU) and, with -m gpu, on the
HIP engine.  Without the variable they are skipped: the build container has no NTL.  The consumer
itself is exercised on a directory of the same layout written from this engine's own objects."""
import os

import pytest

from tests import helib_dump_io as dio

DUMP = os.environ.get("HELIB_DUMP_DIR")


@pytest.mark.parametrize("m,p,bits", [(128, 257, 150), (105, 2, 150)])
def test_the_dump_consumer_on_a_directory_written_by_this_engine(tmp_path, m, p, bits):
    """No HElib here: the files tools/helib_dump/dump.cpp writes, produced instead from this engine's objects through
    helib_amd/wire.py's writers (the reference's binary formats), loaded and replayed by the same code the real dump
    goes through.  Also: a corrupted word in the dumped product is noticed."""
    d = str(tmp_path / "dump")
    dio.make_synthetic_dump(d, m, p, bits)
    loaded = dio.load(d)
    dio.replay(loaded, "oracle")
    idx, rows, h = loaded["prod"]["parts"][1]
    rows = rows.copy()
    rows[0, 5] ^= 1
    loaded["prod"]["parts"][1] = (idx, rows, h)
    with pytest.raises(AssertionError):
        dio.replay(loaded, "oracle")


@pytest.mark.skipif(not DUMP, reason="HELIB_DUMP_DIR not set: needs a directory written by tools/helib_dump (genuine HElib + NTL)")
def test_oracle_replays_the_genuine_helib_dump():
    dio.replay(dio.load(DUMP), "oracle")


@pytest.mark.gpu
@pytest.mark.skipif(not DUMP, reason="HELIB_DUMP_DIR not set: needs a directory written by tools/helib_dump (genuine HElib + NTL)")
def test_hip_engine_replays_the_genuine_helib_dump():
    try:
        import torch  # noqa: F401  (before the library touches the device: tests/test_gpu_parity.py, hx fixture)
    except ImportError:
        pass
    from helib_amd import capi as hx
    if hx.device_count() <= 0:
        pytest.skip("no HIP device")
    dio.replay(dio.load(DUMP), "gpu", hx=hx)
