"""Puts the product package directory (`mix-of-show_amd/`, not a valid Python identifier) on sys.path so
that `import mixofshow` resolves to the MI355X-native implementation with the reference's import paths
(`mixofshow.pipelines.pipeline_edlora`, `mixofshow.models.edlora`, ...)."""
import os
import sys

REPO_ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(REPO_ROOT, 'mix-of-show_amd')
if PKG_DIR not in sys.path:
    sys.path.insert(0, PKG_DIR)
if REPO_ROOT not in sys.path:
    sys.path.insert(0, REPO_ROOT)

# hipGraph replays of the captured training step (TrainEngine.enable_graph): the ROCm 7.2 runtime's "graph packet
# capture" fast path (pre-baked AQL packets + a per-graph kernel-argument pool) faults with
# HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION once ordinary launches are interleaved with replays of a ~4.6 k-node
# graph (reproduced deterministically on MI355X at the 9th optimiser step). Node-by-node graph
# launch is unaffected and costs nothing here (the step stays GPU-bound), so it is selected before the HIP runtime
# initialises. Every entry point imports this module before torch touches the device.
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
