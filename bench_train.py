#!/usr/bin/env python
"""
This file is a shim: the benchmark lives in bench.py (`python bench.py --train ...`).
"""
import bench

if __name__ == '__main__':
    bench.main_train()
