# -*- coding: utf-8 -*-
"""The legs of bench.py (one module per leg) and what they share; bench.py assembles the one JSON line."""
