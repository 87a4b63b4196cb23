import faulthandler, runpy, sys
faulthandler.enable()
faulthandler.dump_traceback_later(int(sys.argv[1]), exit=True)      # a hung launch: print where the main thread sits, then exit
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
