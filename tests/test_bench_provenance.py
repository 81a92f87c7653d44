"""bench.py prices its roofline with PMC figures from profiles/*_pmc_traffic.json only when they were taken on the kernel
sources the run executes: tools/prof_summary.py stamps each file with the SHA-256 of device_math.h +
kernels_trace.h + kernels_shade.h, bench.pmc_profile() compares.  CPU only."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_hash_follows_the_kernel_sources(pt, tmp_path, monkeypatch):
    h = pt.kernel_source_sha256()
    assert len(h) == 64 and h == pt.kernel_source_sha256()
    fake = tmp_path / "pkg"
    shutil.copytree(os.path.join(ROOT, "pathtracer_amd", "csrc"), fake / "csrc", ignore=shutil.ignore_patterns("*.so", "*.o", "host"))
    monkeypatch.setattr(pt, "_HERE", str(fake))
    assert pt.kernel_source_sha256() == h
    with open(fake / "csrc" / "kernels_trace.h", "a") as f:
        f.write("\n// one more comment\n")
    assert pt.kernel_source_sha256() != h   # any edit of the device sources invalidates the committed profiles


def test_a_profile_of_other_sources_is_refused_with_a_reason(pt, tmp_path, monkeypatch):
    import bench
    h = pt.kernel_source_sha256()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    good = {"workload": "bunny-1920x1080-b5", "kernel_sha256": h, "source": "profiles/a_pmc_traffic.json: x", "kernels": {"k_trace": {}}}
    stale = dict(good, kernel_sha256="0" * 64)
    unstamped = {k: v for k, v in good.items() if k != "kernel_sha256"}   # what rounds 1-2 wrote
    (prof / "r01_pmc_traffic.json").write_text(json.dumps(unstamped))
    d, why = bench.pmc_profile("bunny-1920x1080-b5", h)
    assert d is None and "other sources" in why and "r01_pmc_traffic.json" in why
    (prof / "r02_pmc_traffic.json").write_text(json.dumps(good))
    (prof / "r03_pmc_traffic.json").write_text(json.dumps(stale))      # newer by name, wrong sources: skipped, not used
    d, why = bench.pmc_profile("bunny-1920x1080-b5", h)
    assert why is None and d["kernel_sha256"] == h
    d, why = bench.pmc_profile("atrium-1920x1080-b5", h)
    assert d is None and "no PMC profile for workload" in why


def test_the_committed_profile_is_of_the_committed_kernels(pt):
    """The newest profiles/*_pmc_traffic.json of the bench workload carries the hash of the device sources in this tree: whoever
    edits csrc/device_math.h, kernels_trace.h or kernels_shade.h re-runs tools/prof.sh (tools/final_evidence.sh) and commits its
    summary -- otherwise bench.py's line says `frac: null` with the reason, and this test says so first."""
    h = pt.kernel_source_sha256()
    stamped = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        d = json.load(open(f))
        if d.get("workload") == "bunny-1920x1080-b5" and d.get("kernel_sha256"):
            stamped.append((os.path.basename(f), d["kernel_sha256"]))
    assert stamped, "no stamped PMC profile under profiles/"
    assert any(s == h for _, s in stamped), f"kernel sources changed since {stamped[-1][0]} was taken: re-profile (tools/final_evidence.sh)"
