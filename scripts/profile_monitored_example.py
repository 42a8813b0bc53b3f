import cProfile, pstats, sys, os, runpy
sys.argv = ["examples/mnist_monitored.py", "--job_name", "worker", "--task_index", "0", "--protocol", "rccl",
            "--worker_hosts", "127.0.0.1:23456", "--ps_hosts", "127.0.0.1:23457", "--last_step", "3000", "--optimizer", "adam"]
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="23456")
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("examples/mnist_monitored.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
