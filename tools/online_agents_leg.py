"""tools/online_agents_leg.py -- bench_legs.online_agents alone (Python threads + the C++ host tools/online_agents.cpp), JSON on stdout"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs
from dvm_slam_amd import capi, synth
r = bench_legs.online_agents(capi, synth.frame_stream(16), 0)
print(json.dumps(r))
