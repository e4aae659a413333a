#!/bin/bash
MVE_DEBUG=1 timeout 120 python tools/ab_pp2.py 1 small 2>&1 | grep "mve\]" 
