"""CPU oracle: a numpy / torch-CPU restatement of the reference's arithmetic for the hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under agents_amd/ imports this package; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may.  Each module cites the reference
file:line it restates.  The reference itself (tensorflow + tf_agents) cannot be imported in this
environment, so the oracle is pinned against the reference's own known-answer
tests, ported in tests/test_oracle_*.py; what stays unpinned (random index stream, optimizer
arithmetic, initialisers) is listed in DESIGN.md.
"""
