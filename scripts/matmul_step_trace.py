"""sess.run of the N x N fp32 matmul + mean checksum, for a rocprofv3 kernel trace."""
import sys, time
import torch
sys.path.insert(0, ".")
import distributed_tensorflow_amd as tf  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
tf.set_random_seed(1)
A = tf.Variable(tf.random_normal([n, n]))
B = tf.Variable(tf.random_normal([n, n]))
C = tf.matmul(A, B)
checksum = tf.reduce_mean(C)
sess = tf.Session(config=tf.ConfigProto(device="cuda"))
sess.run(tf.global_variables_initializer())
for _ in range(10):
    sess.run(checksum)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    sess.run(checksum)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print("n=%d  %.3f ms/step  %.1f TFLOP/s  stats=%s" % (n, ms, 2.0 * n ** 3 / ms / 1e9, sess.stats))
