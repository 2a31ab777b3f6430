"""The counter passes kept under profiles/ must have been taken from the kernel sources in the tree: bench.py quotes
profiles/traffic_by_size.json when its live passes cannot run, and a profile of other sources says nothing about this library."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_committed_traffic_was_taken_from_these_kernel_sources():
    import bench
    by_size = json.load(open(os.path.join(ROOT, "profiles", "traffic_by_size.json")))
    assert {"3840x2160", "1920x1080"} <= set(by_size)
    sha = bench.kernel_source_sha()
    for size, entry in by_size.items():
        assert entry["kernel_src_sha"] == sha, (
            f"profiles/traffic_by_size.json[{size}] was taken from other kernel sources ({entry['kernel_src_sha']}, tree {sha}): "
            "rerun scripts/regen_profiles.sh and scripts/collect_profiles.sh")
        assert os.path.exists(os.path.join(ROOT, "profiles", entry["source"]))
        w, h = (int(v) for v in size.split("x"))
        assert 0.9 * 9 * w * h < entry["hbm_bytes_per_launch"] < 2.0 * 9 * w * h          # 9 B per pixel are written
    latest = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    assert latest["kernel_src_sha"] == sha


def test_loop_listing_is_of_the_headline_kernel():
    """the kept disassembly names the kernel the bench headline runs and holds the march loop's class counts"""
    txt = open(os.path.join(ROOT, "profiles", "r03_4k_loop_isa.txt")).read()
    assert "rm_queue_kernel4_plain<rot=0, scl=0, 1>" in txt
    assert "valu_packed" in txt and "modelled VALU issue" in txt and "v_rsq_f32" in txt
