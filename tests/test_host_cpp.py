"""The C++20 host interface (namespace jpeg, jpeg-encoder_amd/host): container logic on the CPU, and on the
GPU the whole BaselineEncoder::encode / ::decode path plus the per-block component methods, against the
hashes recorded from the reference and the oracle."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "jpeg-encoder_amd", "host")
GOLDEN = os.path.join(ROOT, "tests", "golden", "reference_hashes.json")


def _binary(name):
    path = os.path.join(HOST, name)
    if not os.path.exists(path):
        subprocess.run(["make", "-s", "-C", HOST], check=True)
    return path


def test_host_containers_cpu(tmp_path):
    r = subprocess.run([_binary("host_cpu_test"), str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all host checks passed" in r.stdout


@pytest.mark.gpu
def test_baseline_encoder_cpp_on_gpu(tmp_path):
    r = subprocess.run([_binary("host_test"), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    golden = {(g["w"], g["h"], g["q"], g["mode"]): g for g in json.load(open(GOLDEN))["rows"]}
    cases = [ln.split() for ln in lines if ln.startswith("case ")]
    assert len(cases) == 7
    for c in cases:
        w, h, q, mode = (int(v) for v in c[1:5])
        g = golden[(w, h, q, mode)]
        assert int(c[6]) == g["jpeg_bytes"], c
        assert c[8] == g["jpeg_fnv"] and c[10] == g["rgb_fnv"], c
        assert c[12] == "1" and (int(c[14]), int(c[15])) == (w, h)
    assert any(ln.startswith("blockwise_encode_matches 1") for ln in lines), r.stdout
    assert "blockwise_decode_matches 1" in lines
    assert "blockwise_decode_eoi_missing" not in r.stdout
    # the reference's other mapper, injected: the GPU path equals the per-block loop
    assert any(ln.startswith("passthrough_encode_matches 1") for ln in lines), r.stdout
    assert "passthrough_decode_matches 1" in lines
    # a user's colour mapper: per-block fallback through the user's virtuals, equal to the library's mapper on the
    # fused path and to the oracle's restatement of that mapper (whose pass-through leg nothing of the reference pins)
    import pyoracle as po
    user = next(ln.split() for ln in lines if ln.startswith("user_mapper_encode_matches"))
    assert user[1] == "1" and int(user[3]) == 6 * 4 and user[5] == "0", r.stdout
    rgb = po.gen_image(45, 30, 1, 3)
    po.set_mapper(1)
    try:
        want = po.encode(rgb, 65)
        want_rgb = po.decode(want, 45, 30, 65)
    finally:
        po.set_mapper(0)
    assert user[7] == po.fnv(want)
    dec = next(ln.split() for ln in lines if ln.startswith("user_mapper_decode_matches"))
    assert dec[1] == "1" and dec[3] == po.fnv(want_rgb.reshape(-1))
    # a user's entropy coder that overrides the protected final-encoding step of the reference's interface
    assert "user_entropy_coder encode_matches 1 decode_matches 1 encoded %d decoded %d" % (3 * 4 * 3, 3 * 4 * 3) in lines, r.stdout
    assert "[Error]: Invalid runtime encoding encountered." in lines and "bad_entries_done" in lines
    # runs are narrowed to 8 bits as the reference narrows them (entropy_encoder.cpp:733): (256, 0) is an end of block
    assert "wide_runs_narrowed encode_matches 1" in lines, r.stdout
    # the injected coder with per-image tables against the oracle's statement of the procedure
    t = next(ln.split() for ln in lines if ln.startswith("tailored_tables bytes"))
    want = po.encode_optimised(po.gen_image(160, 96, 0, 12), 60)
    assert int(t[2]) == len(want) and int(t[2]) < int(t[4]) and t[6] == "1" and t[8] == po.fnv(want), r.stdout
    assert "[Error]: OptimisedHuffmanEncoder codes whole images only (its tables need a pass over all blocks first)" in lines
    assert "tailored_tables_mixed bytes 173" in lines      # the header up to the table segment, as the reference would leave it
    assert "[Error]: Failed to find SOI marker" in lines
    assert "broken_header_pixels 0" in lines
    assert "bmp_round_trip 1" in lines


@pytest.mark.gpu
def test_headless_demo_cli(tmp_path, po):
    """examples/encode_decode: -i/-o/-q/-s flow of the reference demo, files compared with the oracle."""
    import numpy as np

    w, h, q = 120, 64, 60
    rgb = po.gen_image(w, h, 0, 4)
    # 24-bit bottom-up BMP written by hand
    stride = (3 * w + 3) & ~3
    rows = b"".join(rgb[y, :, ::-1].tobytes() + b"\0" * (stride - 3 * w) for y in range(h - 1, -1, -1))
    hdr = (b"BM" + (54 + len(rows)).to_bytes(4, "little") + bytes(4) + (54).to_bytes(4, "little") +
           (40).to_bytes(4, "little") + w.to_bytes(4, "little") + h.to_bytes(4, "little") + (1).to_bytes(2, "little") +
           (24).to_bytes(2, "little") + bytes(4) + len(rows).to_bytes(4, "little") + bytes(16))
    (tmp_path / "in.bmp").write_bytes(hdr + rows)
    out = tmp_path / "out.jpg"
    r = subprocess.run([_binary("encode_decode"), "-i", str(tmp_path / "in.bmp"), "-o", str(out), "-q", str(q), "-s"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    want = po.encode(rgb, q)
    assert out.read_bytes() == want
    dec = (tmp_path / "out.jpg.decoded.bmp").read_bytes()
    want_dec = po.decode(want, w, h, q)
    got = np.frombuffer(dec[54:], np.uint8).reshape(h, stride)[::-1, : 3 * w].reshape(h, w, 3)[:, :, ::-1]
    assert np.array_equal(got, want_dec)
