import json, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import bench_configs as B
for frames in (1, 2):
    for w_out in (820, 585, 455, 390, 341):
        row = {"frames": frames, "ratio": round(4096 / w_out, 1)}
        for label, env in (("k_tile", "1000"), ("two_pass", "1")):
            os.environ["SMOL_BOX2_MIN"] = env
            B.CONFIGS["T"] = (0, 4096, 3072, None, 0, w_out, w_out * 3 // 4, frames, "x")
            row[label] = B.run_config("T", 20, 3, settle=0.1, no_arena=True)["step_us"]
        print(json.dumps(row), flush=True)
