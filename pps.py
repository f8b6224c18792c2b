"""Entry point with the reference's command line (pps.py:75-77): python pps.py {predict,test,rec} -c ... --dotted.overrides"""
from ppsurf_amd.runner import main

if __name__ == '__main__':
    main()
