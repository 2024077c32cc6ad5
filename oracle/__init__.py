"""ORACLE package — CPU restatements of the reference algorithm used ONLY as the checker
(tests/, __graft_entry__.smoke(), bench.py cpu_baseline). Never imported by the product."""
