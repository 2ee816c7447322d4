import os, sys, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["THALLO_MARCH"] = "2"
import test_gpu_distributed as T
if __name__ == "__main__":
    res = T._run(2, 128, 96, 3, 30, True)
    for r in res:
        print(r[0], json.dumps(r[6]), r[7], r[1][:3])
