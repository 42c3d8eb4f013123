#!/usr/bin/env python3
"""The OBJ path at the reference's largest scale, on the CPU (no GPU needed): writes the San-Miguel-class
stand-in (11.0 M triangles) as an OBJ file of the kind the reference loads (v / vt / vn, faces v/vt/vn),
loads it with rtcli --load-only through the multi-threaded and the serial loader, and checks both against
the mesh the file was written from: triangle count, the BvhStats line the library prints for that mesh,
and the mesh fingerprint (equal between the two loaders).

    python scripts/obj_scale.py [--scene san-miguel-class] [--keep]

Prints one JSON line; the file (about 1.3 GB) goes to $TMPDIR and is removed unless --keep is given."""
import argparse
import importlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="san-miguel-class")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--skip-serial", action="store_true")
    args = ap.parse_args()
    rt = importlib.import_module("rust-raytracer_amd")
    scenes = importlib.import_module("rust-raytracer_amd.scenes")
    sc = scenes.load(args.scene)
    path = os.path.join(tempfile.gettempdir(), "%s.obj" % args.scene)
    t0 = time.time()
    scenes.write_obj(sc, path)
    out = {"scene": args.scene, "triangles": int(sc.num_triangles), "file_bytes": os.path.getsize(path),
           "write_s": round(time.time() - t0, 1), "host_threads": os.cpu_count()}
    nodes, order, info, stats = rt.build_bvh_host(sc.positions, sc.indices)
    want = "BvhStats { num_leaves: %d, num_objs: %d, max_depth: %d, max_objs: %d }" % (
        stats["num_leaves"], stats["num_objs"], stats["max_depth"], stats["max_objs"])
    cli = os.path.join(ROOT, "rust-raytracer_amd", "rtcli")
    prints = {}
    for loader in (["parallel"] if args.skip_serial else ["parallel", "serial"]):
        t0 = time.time()
        p = subprocess.run([cli, "--scene", path, "--load-only", "--loader", loader], capture_output=True, text=True)
        dt = time.time() - t0
        assert p.returncode == 0, p.stderr
        assert want in p.stdout, (want, p.stdout)
        m = re.search(r"loaded .*? in ([0-9.]+)(ms|s)", p.stdout)
        fp = re.search(r"mesh fingerprint ([0-9a-f]+)", p.stdout).group(1)
        prints[loader] = fp
        out[loader] = {"wall_s": round(dt, 1), "load_line": (m.group(0) if m else None), "fingerprint": fp}
    if len(prints) == 2:
        assert prints["parallel"] == prints["serial"]
    out["bvh_stats"] = want
    if not args.keep:
        os.remove(path)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
