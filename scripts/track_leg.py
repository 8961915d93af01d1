"""rocprofv3 helper: the tracking leg of bench.py alone (config 2: level-0 iterations of the 2-frame 640x480 tracker in ONE launch of
the persistent level kernel).  python scripts/track_leg.py [iterations]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import tracking_leg  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
print(json.dumps(tracking_leg(torch.device("cuda:0"), steps=steps)))
