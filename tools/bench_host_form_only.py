#!/usr/bin/env python3
"""bench.py's host_form object alone (the drop-in CDemodulator driven with 256-sample calls of host doubles): for traces."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutesdr_amd as ca
import bench
print(json.dumps(bench.host_form(ca, False, check=False)))
