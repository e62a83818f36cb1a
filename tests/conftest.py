import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"),
          os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vectors():
    with open(os.path.join(GOLDEN, "vectors.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def orc():
    import pyoracle
    return pyoracle


def golden_file(name):
    sub = "stream_compressed" if name.endswith((".sz", ".sz-32k", ".sz-64k")) else "data"
    with open(os.path.join(GOLDEN, sub, name), "rb") as fh:
        return fh.read()


DATA_FILES = ["html", "urls.10K", "fireworks.jpeg", "paper-100k.pdf", "html_x_4", "alice29.txt",
              "asyoulik.txt", "lcet10.txt", "plrabn12.txt", "geo.protodata", "kppkn.gtb",
              "Mark.Twain-Tom.Sawyer.txt"]  # tests/test_snappy.nim:93-107
FRAMED_FILES = ["alice29.txt", "house.jpg"] + [f for f in DATA_FILES if f != "alice29.txt"]
