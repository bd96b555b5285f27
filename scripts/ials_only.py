"""The bench's IALS section alone (BASELINE config 5: row-sharded epoch at N = 1 + the emulated 8-way split), one JSON object."""
import json
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = types.SimpleNamespace(no_extras=False)
net = bench.Net(args)
urm = bench.load_urm("ml20m")
extra = {}
bench.ials_section(urm, net, args, extra)
print(json.dumps(extra["ials"]))
