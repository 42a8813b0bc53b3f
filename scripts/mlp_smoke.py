import sys
import numpy as np
import distributed_tensorflow_amd as tf
def build():
    gs = tf.get_variable('global_step', [], initializer=tf.constant_initializer(0), trainable=False)
    x = tf.placeholder(tf.float32, [None, 784]); y_ = tf.placeholder(tf.float32, [None, 10])
    tf.set_random_seed(1)
    with tf.name_scope("weights"):
        W1 = tf.Variable(tf.random_normal([784,100])); W2 = tf.Variable(tf.random_normal([100,10]))
    with tf.name_scope("biases"):
        b1 = tf.Variable(tf.zeros([100])); b2 = tf.Variable(tf.zeros([10]))
    z2 = tf.add(tf.matmul(x,W1),b1); a2 = tf.nn.sigmoid(z2); z3 = tf.add(tf.matmul(a2,W2),b2); y = tf.nn.softmax(z3)
    ce = tf.reduce_mean(-tf.reduce_sum(y_*tf.log(y), axis=[1]))
    train = tf.train.GradientDescentOptimizer(0.0005).minimize(ce, global_step=gs)
    acc = tf.reduce_mean(tf.cast(tf.equal(tf.argmax(y,1), tf.argmax(y_,1)), tf.float32))
    tf.summary.scalar("cost", ce); tf.summary.scalar("accuracy", acc)
    sm = tf.summary.merge_all(); init = tf.global_variables_initializer()
    return locals()
if __name__ == "__main__":
    dev = sys.argv[1] if len(sys.argv) > 1 else "cpu"
    g = tf.Graph()
    with g.as_default():
        m = build()
        rng = np.random.RandomState(0)
        bx = rng.rand(100,784).astype(np.float32)*0.1; by = np.eye(10,dtype=np.float32)[rng.randint(0,10,100)]
        for fuse in (False, True):
            with tf.Session(config=tf.ConfigProto(graph_fusion=fuse, device=dev)) as sess:
                sess.run(m['init'])
                for i in range(4):
                    _, c, s, st = sess.run([m['train'], m['ce'], m['sm'], m['gs']], {m['x']:bx, m['y_']:by})
                    print(fuse, c, st, tf.summary.decode_summary(s))
                print(sess.run(m['acc'], {m['x']:bx,m['y_']:by}), sess.stats)
                plan=[p for p in sess._plans.values()][1]
                print([s.node.type for s in plan.steps])
