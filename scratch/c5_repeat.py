"""C5 leg, time-sliced configuration repeated: is the stand-alone front-end rate of the first configuration representative?"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["DCS_BENCH_C5_NO_SWEEP"] = "1"
import bench, torch
import __graft_entry__ as entry
pkg = entry.load_pkg() if hasattr(entry, "load_pkg") else None
