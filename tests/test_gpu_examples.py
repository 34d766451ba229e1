"""examples/*.py — the reference's preprocess_planar.rs, preprocess_spherical.rs, minimal.rs and spherical.rs on this library — run end to end (small synthesised
sources): the tile counts the reference's examples produce, and a decoded source goes through the same path the parity tests pin."""
import os
import subprocess
import sys

import numpy as np
import pytest

import _oracle as O
import bevy_terrain_amd as bt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)] + list(args), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (script, p.stdout[-1500:], p.stderr[-1500:])
    return p.stdout


def test_planar_example_then_the_view(tmp_path):
    assets = str(tmp_path / "assets")
    out = run("preprocess_planar.py", "--assets", assets, "--size", "1024")
    assert "height: 85 tiles" in out and "albedo: 85 tiles" in out, out
    terrain = os.path.join(assets, "terrains", "planar")
    tiles = bt.tc_decode(open(os.path.join(terrain, "config.tc"), "rb").read())
    assert len(tiles) == 85
    # the height tiles are what the oracle makes of the decoded source file
    src = bt.preprocess.decode_image(os.path.join(terrain, "source", "height.png"), bt.AttachmentFormat.R16)
    oracle = O.OracleAtlas(4, 1024, False, [(512, 2, 1, O.FORMAT_R16)])
    oracle.clear_attachment(0).preprocess_tile(0, src, (0, 4)).run(8)
    for (side, lod, x, y), index in oracle.tiles()[::9]:
        got = np.fromfile(os.path.join(terrain, "data", "height", f"{side}_{lod}_{x}_{y}.bin"), dtype=np.uint16).reshape(512, 512)
        assert np.array_equal(got, oracle.tile(0, index)), (side, lod, x, y)
    out = run("minimal.py", "--assets", assets, "--frames", "24")
    last = [l for l in out.splitlines() if l.startswith("frame")][-1]
    assert int(last.split("final tiles")[1].split()[0]) > 100 and "(0 failed)" in out, out


def test_spherical_example(tmp_path):
    assets = str(tmp_path / "assets")
    out = run("preprocess_spherical.py", "--assets", assets, "--size", "256")
    assert "height: 2046 tiles" in out, out
    assert len(bt.tc_decode(open(os.path.join(assets, "terrains", "spherical", "config.tc"), "rb").read())) == 2046
    out = run("spherical.py", "--assets", assets, "--frames", "24")
    frames = [l for l in out.splitlines() if l.startswith("frame")]
    assert int(frames[-1].split("final tiles")[1].split()[0]) > 100 and "(0 failed)" in out and " 0 tile loads" not in out, out
