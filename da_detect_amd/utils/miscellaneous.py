"""reference: maskrcnn_benchmark/utils/miscellaneous.py:6-11"""
import os


def mkdir(path):
    """create `path` (and parents); an existing directory is not an error"""
    os.makedirs(path, exist_ok=True)
