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
