import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        e=d['extras']
        print(f, "Hvp/s %.0f | b2b %.2f us frac %.3f | hbm %.2f us frac %.3f | spmm %.2f | iter %.1f | hvp in stpcg %.1f | precond %.1f" % (
            d['value'], d['roofline']['kernel_us'], d['roofline']['frac'], d['roofline_hbm']['kernel_us'], d['roofline_hbm']['frac'],
            e.get('spmm_us',0), e.get('stpcg_iteration_us',0), e.get('hvp_in_stpcg_us',0), e.get('preconditioner_apply_us',0)))
    except Exception as ex:
        print(f, "unreadable:", ex)
