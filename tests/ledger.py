"""Parity error ledger (test infrastructure): every GPU parity test records the errors it MEASURED, not only whether it
passed.  Written to gpurun_out/parity_ledger.json on the GPU box (merged back by gpurun); the copy of a full `pytest -m gpu`
run is committed as profiles/rNN_parity.json."""
import json
import os

_ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(_ROOT, "gpurun_out", "parity_ledger.json")


def record(test, **values):
    os.makedirs(os.path.dirname(PATH), exist_ok=True)
    try:
        with open(PATH) as f:
            data = json.load(f)
    except (OSError, ValueError):
        data = {}
    entry = data.setdefault(test, {})
    for k, v in values.items():
        entry[k] = v if isinstance(v, (str, list, dict, int)) else float(v)
    tmp = PATH + ".tmp%d" % os.getpid()
    with open(tmp, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    os.replace(tmp, PATH)
