"""The FF-PWC leg of bench.py on its own (BASELINE configs[3]): one pair eager / hipGraph replay, eight pairs, cost-volume roofline.

    python tools/pwc_leg.py
"""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    out = bench.pwc_measurements(torch.device("cuda:0"))
    cv = out.get("costvolume_roofline", {})
    print(json.dumps({k: v for k, v in out.items() if k != "costvolume_roofline"}, indent=1))
    for lv in cv.get("levels", []):
        print(lv)
    print("dominant", cv.get("dominant_level"))
