for i in 1 2 3; do for e in MLH_TRACK_LOOP=0 MLH_TRACK_LOOP=1; do echo -n "$e  "; env $e python scripts/trackbench.py 2>/dev/null | tail -2 | tr '\n' ' ' | cut -c1-330; echo; done; done
