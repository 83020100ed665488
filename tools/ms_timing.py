#!/usr/bin/env python3
"""Multi-sensor timing: wall time of mom_rt_run_multisensor against the number of sensors (C2-like scene, S points)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import rtamd  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    m = rtamd.scenes.scene_C2(S=S)
    sc = rtamd.prepare_scene(m)
    with rtamd.corert.make_handle(m) as h:
        rtamd.corert.scene_set(h, sc)
        t0 = time.perf_counter(); rtamd.corert.run_scene(h, sc); h.sync(); t_plain = time.perf_counter() - t0
        t0 = time.perf_counter(); rtamd.corert.run_scene(h, sc); h.sync(); t_plain = time.perf_counter() - t0
        print(f"plain rt_run (with the m = 0 reduction): {t_plain * 1e3:.1f} ms")
        for levels in ([20], [10, 30], [5, 15, 25, 35], [4, 12, 20, 28, 36], [0, 5, 10, 15, 20, 25, 30, 35]):
            h.rt_run_multisensor(levels)
            t0 = time.perf_counter(); h.rt_run_multisensor(levels); dt = time.perf_counter() - t0
            print(f"{len(levels)} sensors at {levels}: {dt * 1e3:.1f} ms", flush=True)


if __name__ == "__main__":
    main()
