"""mi_pool's device-resident path without a GPU: the four-phase protocol of gnina_amd/csrc/pool_protocol.h (allocations |
scatter group | scoring | gather group; host-side rendezvous between phases; every rank closes every group it opened; a
failed post aborts the communicators at once; a watchdog names the phase a transport call hung in and aborts) driven on a
mock transport whose group_end blocks until every posted operation is matched -- the property that turns a failing rank
into a hang with the real RCCL.  The reference's fan-out being replaced: gninasrc/lib/parallel_mc.cpp:183-214,
gninasrc/main/main.cpp:1418-1442.  Driver: tests/cpp/test_pool_protocol.cpp (built by gnina_amd.build.build_host)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_protocol_scenarios_on_a_mock_transport():
    exe = os.path.join(ROOT, "gnina_amd", "lib", "test_pool_protocol")
    if not os.path.exists(exe):
        from gnina_amd import build
        build.build_host()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = {l.split()[0]: l for l in r.stdout.splitlines() if l and l.split()[0] in
             ("ok", "ok_empty", "alloc_fails", "score_fails", "send_fails", "hang")}
    assert len(lines) == 6 and all(" PASS " in l for l in lines.values()), r.stdout
    # a rank failing in the scoring phase leaves the transport intact; a failing send abandons it without waiting for the
    # watchdog; the hang is ended by the watchdog, which names the phase
    assert "aborts 0" in lines["score_fails"] and "[alloc > scatter > score]" in lines["score_fails"]
    assert "aborts 1" in lines["send_fails"] and "watchdog" not in lines["send_fails"]
    assert "watchdog" in lines["hang"] and "1: scatter" in lines["hang"]
    assert "mi_pool watchdog: phase '1: scatter" in r.stderr
