import json,sys
d=json.load(open(sys.argv[1]))
print({k:v for k,v in d.items() if k in ("forward_ms","imgs_per_s","frac_of_roof_whole_forward","max_abs_err_vs_oracle_first_images")})
for r in d["layers"][:int(sys.argv[2]) if len(sys.argv)>2 else 20]: print(r["layer"], r["ms"], r["MB"], r["frac_of_hbm_roof"])
