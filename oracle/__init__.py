"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatements of the reference's hot-path arithmetic, used only as checkers by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
Nothing under ``vit-adapter_amd/`` may import this package.
"""
