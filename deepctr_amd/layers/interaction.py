"""Host mirror of the five in-scope layers of the reference's ``deepctr/layers/interaction.py``:
``AFMLayer`` (:39-160), ``CIN`` (:209-341), ``CrossNet`` (:344-435), ``FM`` (:563-607),
``InnerProductLayer`` (:610-694).  Same constructor kwargs, ``get_config`` and weight names/shapes; ``call``
launches the HIP kernels (deepctr_amd/csrc/interaction_kernels.hip, cin_kernels.hip).  The other eleven
interaction layers of the reference are out of scope."""
import torch

from .. import ops
from ..initializers import GlorotNormal, GlorotUniform, Zeros
from .base import Layer


def _stack_fields(inputs):
    """list of F tensors [B,1,E] -> [B,F,E] (the reference concatenates on axis 1 before pairing)."""
    return torch.cat(list(inputs), dim=1)


class AFMLayer(Layer):
    def __init__(self, attention_factor=4, l2_reg_w=0, dropout_rate=0, seed=1024, **kwargs):
        self.attention_factor = attention_factor
        self.l2_reg_w = l2_reg_w
        self.dropout_rate = dropout_rate
        self.seed = seed
        super(AFMLayer, self).__init__(**kwargs)

    def build(self, input_shape):
        if not isinstance(input_shape, list) or len(input_shape) < 2:
            raise ValueError('A `AttentionalFM` layer should be called on a list of at least 2 inputs')
        shape_set = set(tuple(s) for s in input_shape)
        if len(shape_set) > 1:
            raise ValueError('A `AttentionalFM` layer requires inputs with same shapes '
                             'Got different shapes: %s' % (shape_set))
        if len(input_shape[0]) != 3 or input_shape[0][1] != 1:
            raise ValueError('A `AttentionalFM` layer requires inputs of a list with same shape tensor like '
                             '(None, 1, embedding_size)Got different shapes: %s' % (input_shape[0],))
        embedding_size = int(input_shape[0][-1])
        self.add_weight("attention_W", (embedding_size, self.attention_factor), GlorotNormal(seed=self.seed))
        self.add_weight("attention_b", (self.attention_factor,), Zeros())
        self.add_weight("projection_h", (self.attention_factor, 1), GlorotNormal(seed=self.seed))
        self.add_weight("projection_p", (embedding_size, 1), GlorotNormal(seed=self.seed))
        super(AFMLayer, self).build(input_shape)

    def call(self, inputs, training=None, **kwargs):
        if inputs[0].dim() != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % (inputs[0].dim()))
        if training and self.dropout_rate > 0:
            raise NotImplementedError("dropout is a training-time op; the HIP path is inference (forward) only")
        return ops.afm(_stack_fields(inputs), self.w("attention_W"), self.w("attention_b"), self.w("projection_h"),
                       self.w("projection_p"))

    def compute_output_shape(self, input_shape):
        if not isinstance(input_shape, list):
            raise ValueError('A `AFMLayer` layer should be called on a list of inputs.')
        return (None, 1)

    def get_config(self):
        config = {'attention_factor': self.attention_factor, 'l2_reg_w': self.l2_reg_w,
                  'dropout_rate': self.dropout_rate, 'seed': self.seed}
        base = super(AFMLayer, self).get_config()
        base.update(config)
        return base


class CIN(Layer):
    def __init__(self, layer_size=(128, 128), activation='relu', split_half=True, l2_reg=1e-5, seed=1024, **kwargs):
        if len(layer_size) == 0:
            raise ValueError("layer_size must be a list(tuple) of length greater than 1")
        self.layer_size = layer_size
        self.split_half = split_half
        self.activation = activation
        self.l2_reg = l2_reg
        self.seed = seed
        super(CIN, self).__init__(**kwargs)

    def build(self, input_shape):
        if len(input_shape) != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % (len(input_shape)))
        self.build_for(int(input_shape[1]))

    def build_for(self, field_num):
        if self.built:
            return self
        self.field_nums = [int(field_num)]
        for i, size in enumerate(self.layer_size):
            self.add_weight('filter' + str(i), (1, self.field_nums[-1] * self.field_nums[0], size),
                            GlorotUniform(seed=self.seed + i))
            self.add_weight('bias' + str(i), (size,), Zeros())
            if self.split_half:
                if i != len(self.layer_size) - 1 and size % 2 > 0:
                    raise ValueError("layer_size must be even number except for the last layer when split_half=True")
                self.field_nums.append(size // 2)
            else:
                self.field_nums.append(size)
        self.built = True
        return self

    @property
    def filters(self):
        return [self.w('filter%d' % i) for i in range(len(self.layer_size))]

    @property
    def biases(self):
        return [self.w('bias%d' % i) for i in range(len(self.layer_size))]

    def call(self, inputs, **kwargs):
        if inputs.dim() != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % (inputs.dim()))
        return ops.cin(inputs, self.filters, self.biases, list(self.layer_size), self.split_half, self.activation)

    def compute_output_shape(self, input_shape):
        return (None, ops.cin_output_dim(list(self.layer_size), self.split_half))

    def get_config(self):
        config = {'layer_size': self.layer_size, 'split_half': self.split_half, 'activation': self.activation,
                  'seed': self.seed}
        base = super(CIN, self).get_config()
        base.update(config)
        return base


class CrossNet(Layer):
    def __init__(self, layer_num=2, parameterization='vector', l2_reg=0, seed=1024, **kwargs):
        self.layer_num = layer_num
        self.parameterization = parameterization
        self.l2_reg = l2_reg
        self.seed = seed
        print('CrossNet parameterization:', self.parameterization)
        super(CrossNet, self).__init__(**kwargs)

    def build(self, input_shape):
        if len(input_shape) != 2:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 2 dimensions" % (len(input_shape),))
        self.build_for(int(input_shape[-1]))

    def build_for(self, dim):
        if self.built:
            return self
        if self.parameterization == 'vector':
            shape = (dim, 1)
        elif self.parameterization == 'matrix':
            shape = (dim, dim)
        else:
            raise ValueError("parameterization should be 'vector' or 'matrix'")
        for i in range(self.layer_num):
            self.add_weight('kernel' + str(i), shape, GlorotNormal(seed=self.seed))
        for i in range(self.layer_num):
            self.add_weight('bias' + str(i), (dim, 1), Zeros())
        self.dim = dim
        self.built = True
        return self

    def packed(self):
        """kernels [L,d] / [L,d,d] and bias [L,d] as the C ABI wants them (a gather of small tensors)."""
        if self.layer_num == 0:
            return None, None
        ks = torch.stack([self.w('kernel%d' % i).reshape(self.dim, -1) for i in range(self.layer_num)])
        if self.parameterization == 'vector':
            ks = ks.reshape(self.layer_num, self.dim)
        bs = torch.stack([self.w('bias%d' % i).reshape(self.dim) for i in range(self.layer_num)])
        return ks.contiguous(), bs.contiguous()

    def call(self, inputs, **kwargs):
        if inputs.dim() != 2:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 2 dimensions" % (inputs.dim()))
        ks, bs = self.packed()
        return ops.crossnet(inputs, ks, bs, self.parameterization)

    def get_config(self):
        config = {'layer_num': self.layer_num, 'parameterization': self.parameterization, 'l2_reg': self.l2_reg,
                  'seed': self.seed}
        base = super(CrossNet, self).get_config()
        base.update(config)
        return base

    def compute_output_shape(self, input_shape):
        return input_shape


class CrossNetMix(Layer):
    """Mirror of deepctr.layers.interaction.CrossNetMix (:438-560): the cross part of DCN-Mix (mixture of low-rank experts)."""

    def __init__(self, low_rank=32, num_experts=4, layer_num=2, l2_reg=0, seed=1024, **kwargs):
        self.low_rank = low_rank
        self.num_experts = num_experts
        self.layer_num = layer_num
        self.l2_reg = l2_reg
        self.seed = seed
        super(CrossNetMix, self).__init__(**kwargs)

    def build(self, input_shape):
        if len(input_shape) != 2:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 2 dimensions" % (len(input_shape),))
        self.build_for(int(input_shape[-1]))

    def build_for(self, dim):
        if self.built:
            return self
        from .core import Dense
        for k in ('U_list', 'V_list'):
            for i in range(self.layer_num):
                self.add_weight(k + str(i), (self.num_experts, dim, self.low_rank), GlorotNormal(seed=self.seed))
        for i in range(self.layer_num):
            self.add_weight('C_list' + str(i), (self.num_experts, self.low_rank, self.low_rank), GlorotNormal(seed=self.seed))
        # one Dense(1, use_bias=False) per expert, shared by every cross layer (:502, :524)
        self.gating = [Dense(1, use_bias=False, device=self.device).build_for(dim) for _ in range(self.num_experts)]
        self._sublayers.extend(self.gating)
        for i in range(self.layer_num):
            self.add_weight('bias' + str(i), (dim, 1), Zeros())
        self.dim = dim
        self.built = True
        return self

    def packed(self):
        """(U, V [L,experts,d,r], C [L,experts,r,r], gating [experts,d], bias [L,d]) as the C ABI takes them."""
        if self.layer_num == 0:
            return None, None, None, None, None
        st = lambda k: torch.stack([self.w(k + str(i)) for i in range(self.layer_num)]).contiguous()   # noqa: E731
        g = torch.stack([d.w('kernel').reshape(self.dim) for d in self.gating]).contiguous()
        b = torch.stack([self.w('bias%d' % i).reshape(self.dim) for i in range(self.layer_num)]).contiguous()
        return st('U_list'), st('V_list'), st('C_list'), g, b

    def call(self, inputs, **kwargs):
        if inputs.dim() != 2:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 2 dimensions" % (inputs.dim()))
        return ops.crossnet_mix(inputs, *self.packed())

    def get_config(self):
        config = {'low_rank': self.low_rank, 'num_experts': self.num_experts, 'layer_num': self.layer_num,
                  'l2_reg': self.l2_reg, 'seed': self.seed}
        base = super(CrossNetMix, self).get_config()
        base.update(config)
        return base

    def compute_output_shape(self, input_shape):
        return input_shape


class FM(Layer):
    def build(self, input_shape):
        if len(input_shape) != 3:
            raise ValueError("Unexpected inputs dimensions % d, expect to be 3 dimensions" % (len(input_shape)))
        super(FM, self).build(input_shape)

    def call(self, inputs, **kwargs):
        if inputs.dim() != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % (inputs.dim()))
        return ops.fm(inputs)

    def compute_output_shape(self, input_shape):
        return (None, 1)


class BiInteractionPooling(Layer):
    """Mirror of deepctr.layers.interaction.BiInteractionPooling (:170-211)."""

    def build(self, input_shape):
        if len(input_shape) != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % (len(input_shape)))
        super(BiInteractionPooling, self).build(input_shape)

    def call(self, inputs, **kwargs):
        if inputs.dim() != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % (inputs.dim()))
        return ops.bi_interaction(inputs)

    def compute_output_shape(self, input_shape):
        return (None, 1, input_shape[-1])


class OutterProductLayer(Layer):
    """Weight holder only: PNN (reference models/pnn.py:49) instantiates this layer, and thereby its kernel, even with
    use_outter=False, so Keras weight lists of a reference PNN contain it.  The outer-product arithmetic
    (interaction.py:866-924) is outside SURVEY §8 and calling the layer raises."""

    def __init__(self, kernel_type='mat', seed=1024, **kwargs):
        if kernel_type not in ['mat', 'vec', 'num']:
            raise ValueError("kernel_type must be mat,vec or num")
        self.kernel_type = kernel_type
        self.seed = seed
        super(OutterProductLayer, self).__init__(**kwargs)

    def build_for(self, num_inputs, embed_size):
        if self.built:
            return self
        num_pairs = int(num_inputs * (num_inputs - 1) / 2)
        shape = {'mat': (embed_size, num_pairs, embed_size), 'vec': (num_pairs, embed_size), 'num': (num_pairs, 1)}[self.kernel_type]
        self.add_weight('kernel', shape, GlorotUniform(seed=self.seed))
        self.built = True
        return self

    def call(self, inputs, **kwargs):
        raise NotImplementedError("OutterProductLayer arithmetic is outside the MI355X hot-path scope (SURVEY.md §8)")

    def get_config(self):
        base = super(OutterProductLayer, self).get_config()
        base.update({'kernel_type': self.kernel_type, 'seed': self.seed})
        return base


class InnerProductLayer(Layer):
    def __init__(self, reduce_sum=True, **kwargs):
        self.reduce_sum = reduce_sum
        super(InnerProductLayer, self).__init__(**kwargs)

    def build(self, input_shape):
        if not isinstance(input_shape, list) or len(input_shape) < 2:
            raise ValueError('A `InnerProductLayer` layer should be called on a list of at least 2 inputs')
        shape_set = set(tuple(s) for s in input_shape)
        if len(shape_set) > 1:
            raise ValueError('A `InnerProductLayer` layer requires inputs with same shapes '
                             'Got different shapes: %s' % (shape_set))
        if len(input_shape[0]) != 3 or input_shape[0][1] != 1:
            raise ValueError('A `InnerProductLayer` layer requires inputs of a list with same shape tensor like '
                             '(None,1,embedding_size)Got different shapes: %s' % (input_shape[0],))
        super(InnerProductLayer, self).build(input_shape)

    def call(self, inputs, **kwargs):
        if inputs[0].dim() != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % (inputs[0].dim()))
        return ops.inner_product(_stack_fields(inputs), self.reduce_sum)

    def compute_output_shape(self, input_shape):
        num_inputs = len(input_shape)
        num_pairs = int(num_inputs * (num_inputs - 1) / 2)
        input_shape = input_shape[0]
        if self.reduce_sum:
            return (input_shape[0], num_pairs, 1)
        return (input_shape[0], num_pairs, input_shape[-1])

    def get_config(self):
        config = {'reduce_sum': self.reduce_sum}
        base = super(InnerProductLayer, self).get_config()
        base.update(config)
        return base
