"""Mirrors the reference's `lib` package for the inference hot path (reference lib/__init__.py:112-113)."""
import os

BASE_DIR = os.getcwd()
BASE_MODELS_DIR = os.path.join(BASE_DIR, "models")
