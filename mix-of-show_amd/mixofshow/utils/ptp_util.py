"""AttentionControl / AttentionStore — the controller protocol of the reference's
mixofshow/utils/ptp_util.py:22-108 on the fused attention path.

The reference stores every cross-attention probability map (B*H, N, 77) of a UNet forward and
`cal_attn_reg` (trainer_edlora.py:263-313) then reads only the head-mean of the columns at the concept
token positions. Here the controller DECLARES those positions (`set_token_positions`) and receives
(B, H, N, T) tensors holding exactly those columns, produced inside the attention kernel (with autograd
through them). Bookkeeping (cur_att_layer / cur_step / between_steps / get_average_attention / reset)
follows the reference line by line in behaviour. The notebook visualisation helpers (:111-200) are
out of scope (cv2 / IPython).
"""
import abc


class EmptyControl:
    """reference ptp_util.py:11-19: the no-op controller. `is_passthrough` tells the processor that it neither reads nor edits
    the map, so no (B*H, N, 77) tensor is materialised for it and the sampling loop may replay a captured graph."""
    is_passthrough = True

    def step_callback(self, x_t):
        return x_t

    def between_steps(self):
        return

    def __call__(self, attn, is_cross, place_in_unet):
        return attn


class AttentionControl(abc.ABC):

    def __init__(self, low_resource, training):
        self.cur_step = 0
        self.num_att_layers = -1
        self.cur_att_layer = 0
        self.low_resource = low_resource
        self.training = training
        self.token_positions = None

    def set_token_positions(self, tok_idx):
        """tok_idx: int32 device tensor (B, T), T <= 4 — key positions whose probabilities are consumed."""
        self.token_positions = tok_idx

    def step_callback(self, x_t):
        return x_t

    def between_steps(self):
        return

    @property
    def num_uncond_att_layers(self):
        return self.num_att_layers if self.low_resource else 0

    @abc.abstractmethod
    def forward(self, attn, is_cross, place_in_unet):
        raise NotImplementedError

    def __call__(self, attn, is_cross, place_in_unet):
        if self.cur_att_layer >= self.num_uncond_att_layers:
            if self.low_resource or self.training:
                attn = self.forward(attn, is_cross, place_in_unet)
            else:
                # reference ptp_util.py:45-46: in eval only the conditional half of the CFG batch is edited
                h = attn.shape[0]
                attn[h // 2:] = self.forward(attn[h // 2:], is_cross, place_in_unet)
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers + self.num_uncond_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            self.between_steps()
        return attn

    def reset(self):
        self.cur_step = 0
        self.cur_att_layer = 0


class AttentionStore(AttentionControl):

    @staticmethod
    def get_empty_store():
        return {k: [] for k in ('down_cross', 'mid_cross', 'up_cross', 'down_self', 'mid_self', 'up_self')}

    def __init__(self, low_resource=False, training=False):
        super().__init__(low_resource, training)
        self.step_store = self.get_empty_store()
        self.attention_store = {}

    def forward(self, attn, is_cross, place_in_unet):
        self.step_store[f"{place_in_unet}_{'cross' if is_cross else 'self'}"].append(attn)
        return attn

    def between_steps(self):
        if len(self.attention_store) == 0:
            self.attention_store = self.step_store
        else:
            for key in self.attention_store:
                for i in range(len(self.attention_store[key])):
                    self.attention_store[key][i] = self.attention_store[key][i] + self.step_store[key][i]
        self.step_store = self.get_empty_store()

    def get_average_attention(self):
        if self.cur_step == 1:      # x / 1 == x exactly: the training loop resets the store every step (16 + 16 launches saved)
            return {key: list(self.attention_store[key]) for key in self.attention_store}
        return {key: [item / self.cur_step for item in self.attention_store[key]] for key in self.attention_store}

    def reset(self):
        super().reset()
        self.step_store = self.get_empty_store()
        self.attention_store = {}
