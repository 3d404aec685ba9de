import torch, runpy, sys
sys.argv = ["tools/host_api_latency.py"]
runpy.run_path("tools/host_api_latency.py", run_name="__main__")
