import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = open(os.path.join(ROOT, "tools", "pp_train_check2.py")).read().split("CHILD = r'''")[1].split("''' % ROOT")[0] % ROOT
CHILD = "import os\n" + CHILD
for lib in sys.argv[1:]:
    print("lib", lib, flush=True)
    subprocess.run([sys.executable, "-c", CHILD, "train"], env=dict(os.environ, PNR_MLP_VARIANT="2", PNR_LIB_PATH=os.path.join(ROOT, "build", "ab", "libpnr_%s.so" % lib)), timeout=120)
