"""TEST INFRASTRUCTURE - not a product path.

`oracle/` holds CPU restatements of the reference algorithm for the camera-to-BEV hot path, used
only as the checker: by `tests/`, by `__graft_entry__.smoke()` and by `bench.py`'s `cpu_baseline`
leg.  Nothing under `fiery_amd/` imports it.
"""
