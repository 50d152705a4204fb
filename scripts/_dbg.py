import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); os.chdir("/root/repo")
import test_gpu_mesh as TMe
for cfg in ((113, 2.1731277655695598), (101, 3.9018032325839442), (41, 2.916409648779724), (89, 1.8475194026371131)):
    try:
        TMe.test_forward_and_backward_match_oracle(*cfg); print("mesh", cfg, "ok")
    except AssertionError as e:
        print("mesh", cfg, "FAIL", repr(e)[:200])
